"""GPU probe: GPU time of one WSJ-base training step attributed to C-ABI calls (HIP events around every call)."""
import os, sys, collections
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
lib = rec.lib
orig = lib.call
events = []
def timed(name, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    orig(name, *args)
    e1.record(torch.cuda.current_stream())
    events.append((name, e0, e1))
lib.call = timed
s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with rec._on_stream():
    s0.record(torch.cuda.current_stream())
tr.train_step(batch)
with rec._on_stream():
    s1.record(torch.cuda.current_stream())
torch.cuda.synchronize()
lib.call = orig
tot = collections.OrderedDict()
for i, (name, e0, e1) in enumerate(events):
    t = tot.setdefault(name, [0, 0.0])
    t[0] += 1; t[1] += e0.elapsed_time(e1)
print("whole step (events): %.2f ms" % s0.elapsed_time(s1))
acc = 0.0
for k, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    acc += ms
    print("  %-28s %5d calls %8.3f ms" % (k, n, ms))
print("  sum inside calls %.2f ms; between calls (torch ops, launch gaps) %.2f ms" % (acc, s0.elapsed_time(s1) - acc))
# per-call detail for the recurrent layers
for name, e0, e1 in events:
    if name.startswith("lvsr_bigru") or name.startswith("lvsr_attdec_"):
        print("    %-24s %.3f ms" % (name, e0.elapsed_time(e1)))
