"""GPU probe: replay each real encoder-layer graph of a WSJ-base recognizer in isolation (sync before, 3 replays)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import torch
from lvsr_amd import spec, synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
factory, B, T, L = spec.WORKLOADS["wsj_base"]
cfg = factory()
dev = torch.device("cuda:0")
rec = SpeechRecognizer(device=dev, params=synthetic.make_params(cfg, seed=10), net_config=cfg)
batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(cfg, B, T, L, seed=1234).items()}
for _ in range(2):
    rec.cost_and_gradients(batch)
torch.cuda.synchronize()
lib = rec.lib
orig = lib.run
def run(fn, struct_name, ref, use_graph=None, **fields):
    out = orig(fn, struct_name, ref, use_graph, **fields)
    if fn in ("lvsr_bigru_fwd", "lvsr_bigru_bwd"):
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream()); orig(fn, struct_name, ref, use_graph, **fields); e1.record(torch.cuda.current_stream())
            e1.synchronize(); ts.append(e0.elapsed_time(e1))
        Tn = fields["T"]
        print("%s T=%d sub=%d mask=%s: %s ms -> %.2f us per step" % (fn, Tn, fields["sub"], fields["mask"] is not None,
              ", ".join("%.3f" % t for t in ts), min(ts) * 1e3 / Tn), flush=True)
    return out
lib.run = run
rec.cost_and_gradients(batch)
torch.cuda.synchronize()
