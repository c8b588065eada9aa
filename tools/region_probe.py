"""GPU probe: is the whole-step graph region captured and replayed, and what does a step cost with / without it?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import torch
from lvsr_amd import spec, synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
from lvsr_amd.training import Trainer
dev = torch.device("cuda:0")
factory, B, T, L = spec.WORKLOADS["wsj_base"]
cfg = factory()
rec = SpeechRecognizer(device=dev, params=synthetic.make_params(cfg, seed=10), net_config=cfg)
tr = Trainer(rec, gradient_threshold=100.0, rules=("momentum", "adadelta"), scale=0.1, decay_rate=0.95, epsilon=1e-8, max_norm=1.0)
batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(cfg, B, T, L, seed=1).items()}
for s in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cm = tr.train_step(batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    st = [(v["seen"], v.get("bad")) for v in rec._regions.values()]
    print("step %d: %.2f ms  cost %.3f  regions %s  graphs %d  ws.generation %d" % (s, dt, float(cm.sum()) / B, st, rec.lib._lvsr_graph_count(), rec.ws.generation), flush=True)
