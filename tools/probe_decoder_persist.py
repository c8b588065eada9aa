"""GPU probe: the attention decoder's label loop (generator.cost_matrix forward) on a workload's shapes, step kernels vs the
persistent cluster kernel (csrc/decoder_persist.hip), and the persistent kernel's phase clock (work-group 0).
    python tools/probe_decoder_persist.py [workload] [prior]
"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import numpy, torch
from lvsr_amd import spec, synthetic
from lvsr_amd import native
from lvsr_amd.bricks.recognizer import SpeechRecognizer

KNOBS = [a for a in sys.argv[1:] if "=" in a]          # e.g. dec_cluster=8 (include/lvsr_hip.h LVSR_KNOB_*)
sys.argv = [a for a in sys.argv if "=" not in a]
native.get().set_knobs(KNOBS)
name = sys.argv[1] if len(sys.argv) > 1 else "wsj_base"
prior = sys.argv[2] if len(sys.argv) > 2 else None
factory, B, T, L = spec.WORKLOADS[name]
cfg = factory()
if prior == "median":
    cfg["prior"] = dict(type="window_around_median", before=20, after=60)
params = synthetic.make_params(cfg, seed=1)
batch = synthetic.make_batch(cfg, B, T, L, seed=2, ragged=False)
PH = ["S gather", "A: sW/sg dots + publish", "SW gather", "B: energies", "EN gather", "C: softmax", "D: gate sums + publish",
      "conv (SW shadow)", "centre scan (RS shadow)", "centres + window", "RS gather", "E: candidate", "  conv: operands + MFMA", "  conv: partials + barrier", "  conv: fold + store"]
ref = None
for mode, prof in (("0", "0"), ("1", "0"), ("1", "1")):
    native.get().set_knob("phase_clock", int(prof))
    rec = SpeechRecognizer(device="cuda:0", params=params, net_config=cfg, use_persistent_decoder=(mode == "1"))
    gen = rec.generator
    x = torch.from_numpy(batch["recordings"]).cuda(); xm = torch.from_numpy(batch["recordings_mask"]).cuda()
    y = torch.from_numpy(batch["labels"]).cuda(); ym = torch.from_numpy(batch["labels_mask"]).cuda()
    enc, em = rec.encoder.apply(x, xm)
    best = 1e9
    for it in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        cm = gen.cost_matrix(y, ym, enc, em)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    gen.check_persistent()
    tot = float(cm.sum())
    print("%s decoder cost_matrix forward, %s: %.3f ms (%.2f us/label), cost sum %.6f" % (
        name, ("persistent + phase clock" if prof == "1" else "persistent") if mode == "1" else "step kernels", best, best * 1e3 / L, tot), flush=True)
    if prof == "1":
        sync = [b for k, b in gen.ws._bufs.items() if k[0] == "gen.sync"][0]
        clk = sync[16:16 + 2 * len(PH)].cpu().numpy().view(numpy.int64)
        for nm, c in zip(PH, clk):
            print("    %-28s %7.3f us/label" % (nm, c * 0.01 / L))
        print("    %-28s %7.3f us/label" % ("sum", clk.sum() * 0.01 / L))
        print("    rel. cost difference vs step kernels %.2e" % (abs(tot - ref) / abs(ref)))
    if mode == "0":
        ref = tot
