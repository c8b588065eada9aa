"""GPU probe: per-phase host-enqueue time vs GPU time of one WSJ-base training step (sync between phases)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import torch
from lvsr_amd import spec, synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
from lvsr_amd.training import Trainer
import bench

factory, B, T, L = spec.WORKLOADS["wsj_base"]
cfg = factory()
dev = torch.device("cuda:0")
rec = SpeechRecognizer(device=dev, params=synthetic.make_params(cfg, seed=10), net_config=cfg)
tr = Trainer(rec, distributed=False, **bench.TRAIN_CONF)
batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(cfg, B, T, L, seed=1234).items()}
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()

def phase(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with rec._on_stream():
        out = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-28s host enqueue %7.2f ms   until GPU done %7.2f ms" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
    return out

for rep in range(2):
    x = rec._t(batch["recordings"], torch.float32, "recordings"); xm = rec._t(batch["recordings_mask"], torch.float32, "recordings_mask")
    y = rec._t(batch["labels"], torch.int64, "labels"); ym = rec._t(batch["labels_mask"], torch.float32, "labels_mask")
    enc = phase("encoder forward", lambda: rec.encoder.apply(x, xm))
    phase("generator cost_matrix", lambda: rec.generator.cost_matrix(y, ym, attended=enc[0], attended_mask=enc[1]))
    dA = phase("generator backward", lambda: rec.generator.backward())
    phase("encoder backward", lambda: rec.encoder.backward(dA))
    phase("optimiser", lambda: tr.apply_gradients(B))
    t0 = time.perf_counter(); tr.train_step(batch); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("whole step: host %.2f ms, total %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
