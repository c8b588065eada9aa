"""GPU probe: the phase clock of the persistent decoder's reverse walk (csrc/decoder_persist_bwd.hip) kept by EVERY work-group of the first cluster
in turn (knob phase_clock = 1 + p): which work-group is the laggard of which exchange.
    python tools/probe_decoder_bwd_skew.py [workload] [knob=value ...]
"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import numpy, torch
from lvsr_amd import native, spec, synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer

KNOBS = [a for a in sys.argv[1:] if "=" in a]
sys.argv = [a for a in sys.argv if "=" not in a]
native.get().set_knobs(KNOBS)
name = sys.argv[1] if len(sys.argv) > 1 else "wsj_base"
factory, B, T, L = spec.WORKLOADS[name]
cfg = factory()
params = synthetic.make_params(cfg, seed=1)
batch = synthetic.make_batch(cfg, B, T, L, seed=2, ragged=False)
PH = ["loads+dpc+F", "A gather", "drh,pubB", "B gather", "q publish", "C gat+de", "energies", "dsW+D gat", "WsT+E pub", "correl", "E gat+ds", "dsacc dots", "q contr"]
rows = []
P = None
for p in range(32):
    native.get().set_knob("phase_clock", 1 + p)
    rec = SpeechRecognizer(device="cuda:0", params=params, net_config=cfg)
    gen = rec.generator
    x = torch.from_numpy(batch["recordings"]).cuda(); xm = torch.from_numpy(batch["recordings_mask"]).cuda()
    y = torch.from_numpy(batch["labels"]).cuda(); ym = torch.from_numpy(batch["labels_mask"]).cuda()
    enc, em = rec.encoder.apply(x, xm)
    for it in range(2):
        gen.cost_matrix(y, ym, enc, em)
        gen.backward()
    torch.cuda.synchronize()
    gen.check_persistent()
    if P is None:
        P = -(-cfg["dim_dec"] // 16)          # clusters of ceil(D / 16) work-groups (the default shape)
    sync = [b for k, b in gen.ws._bufs.items() if k[0] == "gen.sync_bwd"][0]
    clk = sync[16:16 + 2 * len(PH)].cpu().numpy().view(numpy.int64)
    rows.append(clk * 0.01 / L)
    if p + 1 >= P:
        break
print("%s reverse walk: us per label by phase (columns) and work-group of the first cluster (rows)" % name)
print("  p  " + " ".join("%10s" % h for h in PH) + "        sum")
for p, r in enumerate(rows):
    print("%3d  " % p + " ".join("%10.3f" % v for v in r) + " %10.3f" % r.sum())
