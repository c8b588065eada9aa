"""GPU probe: the attention decoder's backward (generator.backward) on a workload's shapes, step kernels vs the persistent
kernel (csrc/decoder_persist_bwd.hip), and the persistent kernel's phase clock (work-group 0).
    python tools/probe_decoder_persist_bwd.py [workload]
"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import numpy, torch
from lvsr_amd import native
from lvsr_amd import spec, synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer

KNOBS = [a for a in sys.argv[1:] if "=" in a]          # e.g. dec_cluster=8 (include/lvsr_hip.h LVSR_KNOB_*)
sys.argv = [a for a in sys.argv if "=" not in a]
native.get().set_knobs(KNOBS)
name = sys.argv[1] if len(sys.argv) > 1 else "wsj_base"
factory, B, T, L = spec.WORKLOADS[name]
cfg = factory()
params = synthetic.make_params(cfg, seed=1)
batch = synthetic.make_batch(cfg, B, T, L, seed=2, ragged=False)
PH = ["loads + publish dpc + alignment-gradient gather", "A gather", "drh, publish B, AW loads", "B gather", "dsacc + q + publish",
      "C gather + sd + de", "energies backward", "dsW publish + D gather", "Ws^T part + E publish", "alignment correlation", "E gather + ds", "  (dsacc dots)", "  (q contraction)"]
from lvsr_amd import native
# (persistent?, phase clock?): the step kernels, then the persistent kernel without / with the phase clock of work-group 0
for mode, prof in (("0", "0"), ("1", "0"), ("1", "1")):
    native.get().set_knob("phase_clock", int(prof))
    rec = SpeechRecognizer(device="cuda:0", params=params, net_config=cfg)
    gen = rec.generator
    gen.use_persistent_bwd = mode == "1"          # "0": the four step kernels per label for the reverse walk
    x = torch.from_numpy(batch["recordings"]).cuda(); xm = torch.from_numpy(batch["recordings_mask"]).cuda()
    y = torch.from_numpy(batch["labels"]).cuda(); ym = torch.from_numpy(batch["labels_mask"]).cuda()
    enc, em = rec.encoder.apply(x, xm)
    best = 1e9
    for it in range(4):
        cm = gen.cost_matrix(y, ym, enc, em)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gen.backward()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    gen.check_persistent()
    what = "step kernels" if mode == "0" else "persistent%s" % (" + phase clock" if prof == "1" else "")
    print("%s generator.backward, %s: %.3f ms (%.2f us/label)" % (name, what, best, best * 1e3 / L), flush=True)
    if prof == "1":
        sync = [b for k, b in gen.ws._bufs.items() if k[0] == "gen.sync_bwd"][0]
        clk = sync[16:16 + 2 * len(PH)].cpu().numpy().view(numpy.int64)
        for nm, c in zip(PH, clk):
            print("    %-48s %7.3f us/label" % (nm, c * 0.01 / L))
        print("    %-48s %7.3f us/label" % ("sum", clk.sum() * 0.01 / L))
