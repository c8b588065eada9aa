"""GPU probe: persistent decoder vs step kernels on the ragged full-size batch with a window_around_median prior: where do the
costs differ (per utterance / label), do window centres differ?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import numpy, torch
from lvsr_amd import spec, synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
cfg = spec.wsj_base()
cfg["prior"] = dict(type="window_around_median", before=20, after=60)
params = synthetic.make_params(cfg, seed=10)
batch = synthetic.make_batch(cfg, 16, 800, 100, seed=77, ragged=True)
out = {}
for persistent in (True, False):
    rec = SpeechRecognizer(device="cuda:0", params=params, net_config=cfg, use_persistent_decoder=persistent)
    cm = rec.cost_and_gradients(batch).cpu().numpy()
    torch.cuda.synchronize()
    pos = [b for k, b in rec.ws._bufs.items() if k[0] == "gen.pos"][0].cpu().numpy().copy().reshape(-1)[:101 * 16].reshape(101, 16)
    out[persistent] = (cm, pos, rec.generator.last["weights"].cpu().numpy().copy(), rec.generator.last["energies"].cpu().numpy().copy())
cp, pp, wp, ep = out[True]; cs, ps, ws, es = out[False]
print("cost sums", cp.sum(), cs.sum())
d = numpy.abs(cp - cs)
print("per-utterance max |dcost|:", numpy.round(d.max(0), 5))
print("labels with |dcost| > 1e-3:", numpy.argwhere(d > 1e-3)[:20].tolist())
L = cp.shape[0]
dp = numpy.abs(pp[:L] - ps[:L])
print("window centres differ at (label, utt):", numpy.argwhere(dp > 0)[:20].tolist())
for (l, b) in numpy.argwhere(dp > 0)[:5]:
    print("  label %d utt %d: centre %.1f vs %.1f; cumsum near 0.5: %s" % (l, b, pp[l, b], ps[l, b],
          numpy.round(numpy.cumsum(wp[l - 1, b])[int(min(pp[l, b], ps[l, b])) - 1:int(max(pp[l, b], ps[l, b])) + 2], 6).tolist()))
print("max |dW|", numpy.abs(wp - ws).max(), "max |dE|", numpy.abs(ep - es).max())
first = numpy.argwhere(numpy.abs(wp - ws) > 1e-4)
print("first alignment difference > 1e-4 at (label, utt, t):", first[:3].tolist())
l, b, t = [int(v) for v in first[0]]
am = batch["recordings_mask"]
print("encoded mask length of utt %d:" % b, int(rec.encoded_mask[:, b].sum()) if hasattr(rec, "encoded_mask") else "?")
print("label %d utt %d: centre(slot l) %.1f  centre(slot l+1) %.1f" % (l, b, pp[l, b], pp[l + 1, b]))
print("all centres at slot l:", pp[l].tolist())
print("E persistent", numpy.round(ep[l, b, t - 4:t + 5], 4).tolist())
print("E step      ", numpy.round(es[l, b, t - 4:t + 5], 4).tolist())
print("W persistent", numpy.round(wp[l, b, t - 4:t + 5], 5).tolist())
print("W step      ", numpy.round(ws[l, b, t - 4:t + 5], 5).tolist())
de = numpy.abs(ep - es)
fe = numpy.argwhere(de > 1e-3)
print("first energy difference > 1e-3 at (label, utt, t):", fe[:5].tolist(), "values", [(float(ep[tuple(x)]), float(es[tuple(x)])) for x in fe[:3]])
nz_p = numpy.flatnonzero(ep[l, b]); nz_s = numpy.flatnonzero(es[l, b])
print("energy support persistent [%d,%d] step [%d,%d]" % (nz_p.min(), nz_p.max(), nz_s.min(), nz_s.max()))
print("max |dE| per label (all utterances), labels 0..15:", ["%.1e" % de[x].max() for x in range(16)])
print("max |dcost| per label, labels 0..15:", ["%.1e" % d[x].max() for x in range(16)])
