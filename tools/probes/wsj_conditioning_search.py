"""Search per-group parameter scales on which the float32 and float64 ORACLES agree at full WSJ size under
window_around_median(10, 100) (round-3 verdict, next-round item 1).

    python tools/probes/wsj_conditioning_search.py train  '{"transition.state_to": 0.4, ...}' [B] [seed] [scale]
    python tools/probes/wsj_conditioning_search.py decode '{...}' [utt] [seed] [scale]

`train`: teacher-forced cost of the WSJ-base network on a (B, 800 frames, 100 labels) batch in both precisions: relative
difference of the summed cost, worst per-label cost difference, fraction of equal alignment argmax, fraction of equal window
centres (median), peakedness of the alignments, and how far the alignment travels (a fixture whose attention never moves
pins nothing).  `decode`: beam 16 + FST LM with exp/wsj/decode.sh settings in both precisions: length of the identical
ranked head.  The scales found are recorded in oracle/theano_harness/gen_golden.py (WSJ_COND).
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "attention-lvcsr_amd")):
    sys.path.insert(0, p)
import numpy
import torch
from lvsr_amd import spec, synthetic
from oracle import lvsr_oracle as O, lm_oracle as LO


def median_pos(w):
    """window_around_median's position (lvsr/bricks/attention.py:138-144) of alignments (L, B, T')."""
    cs = numpy.cumsum(w.astype(numpy.float64), axis=2) - 0.5 >= 0
    return numpy.argmax(numpy.diff(cs.astype(numpy.int8), axis=2), axis=2)


def train(scales, B, seed, scale):
    cfg = spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100))
    params = synthetic.make_params(cfg, seed=seed, scale=scale, scales=scales)
    batch = synthetic.make_batch(cfg, B, 800, 100, seed=1234)
    res, grads = {}, {}
    for dt in (torch.float32, torch.float64):
        orc = O.OracleRecognizer(cfg, params, dtype=dt)
        if os.environ.get("GRAD", "1") == "1":          # gradients too: the backward chain has a conditioning of its own
            out, grads[dt] = orc.cost_and_grads(batch)
            out = {k: v.detach() for k, v in out.items()}
        else:
            with torch.no_grad():
                out = orc.cost(batch["recordings"], batch["recordings_mask"], batch["labels"], batch["labels_mask"])
        res[dt] = (out["cost_matrix"].numpy().astype(numpy.float64), out["weights"].numpy())
    a, b = res[torch.float32], res[torch.float64]
    if grads:
        worst = ("", 0.0)
        for k, g64 in grads[torch.float64].items():
            g32 = grads[torch.float32][k]
            rel = float(numpy.abs(g32 - g64).max() / max(1e-3, numpy.abs(g64).max()))
            if rel > worst[1]:
                worst = (k, rel)
        print("  gradients float32 vs float64: worst max-abs difference relative to the tensor's max %.2e (%s)" % (worst[1], worst[0][-60:]))
    d = numpy.abs(a[0] - b[0])
    am = b[1].argmax(2)
    print(json.dumps(scales), "seed", seed, "scale", scale, "B", B)
    print("  cost sum %.4f rel diff %.2e; worst label diff %.2e (rel %.2e)" % (
        b[0].sum(), abs(a[0].sum() - b[0].sum()) / b[0].sum(), d.max(), (d / numpy.abs(b[0])).max()))
    print("  argmax equal %.4f  median equal %.4f  max weight mean %.3f min %.3f" % (
        (a[1].argmax(2) == am).mean(), (median_pos(a[1]) == median_pos(b[1])).mean(), b[1].max(axis=2).mean(), b[1].max(axis=2).min()))
    # margin of the median: how far cumsum-0.5 is from crossing at the neighbours (a step function: small margins flip)
    cs = numpy.cumsum(b[1].astype(numpy.float64), axis=2) - 0.5
    print("  median margin: min |cumsum - 0.5| %.2e; argmax margin min (top1-top2) %.2e" % (
        numpy.abs(cs).min(), numpy.min(numpy.sort(b[1], axis=2)[:, :, -1] - numpy.sort(b[1], axis=2)[:, :, -2])))
    print("  argmax path utt0:", am[::8, 0].tolist(), " distinct positions per utt:", [len(set(am[:, u].tolist())) for u in range(B)])
    print("  cost per label utt0:", numpy.round(b[0][::10, 0], 3).tolist())


def lm_arcs(V, fst_seed):
    import math
    rng = numpy.random.RandomState(fst_seed)
    arcs, backoff = [], V + 1
    uni = rng.dirichlet(numpy.ones(V) * 2.0)
    for s_ in [0] + list(range(1, V + 1)):
        keep = rng.choice(V, size=max(2, V // 2), replace=False)
        pr = rng.dirichlet(numpy.ones(len(keep)))
        for c, pc in zip(keep, pr):
            arcs.append((s_, 1 + int(c), int(c) + 1, -math.log(0.8 * pc)))
        arcs.append((s_, backoff, 0, -math.log(0.2)))
    for c in range(V):
        arcs.append((backoff, 1 + c, c + 1, -math.log(uni[c])))
    return arcs


def decode(scales, utt, seed, scale):
    cfg = spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100))
    cfg["max_decoded_length_scale"] = 3.0
    V = cfg["num_phonemes"]
    params = synthetic.make_params(cfg, seed=seed, scale=scale, scales=scales)
    lm = dict(dense=LO.DenseFST(lm_arcs(V, 9), 0, V), remap={c: c + 1 for c in range(V)}, no_transition_cost=20.0, weight=0.5)
    x = numpy.random.RandomState(100 + utt).normal(size=(800, cfg["input_dim"])).astype("float32")
    res = {}
    for dt in (torch.float32, torch.float64):
        t0 = time.time()
        orc = O.OracleRecognizer(cfg, params, dtype=dt)
        try:
            res[dt] = orc.beam_search(x, 16, char_discount=1.0, round_to_inf=1e9, stop_on="optimistic_future_cost", lm=lm)
        except LookupError:
            res[dt] = ([], [])
        print(dt, "%.0fs" % (time.time() - t0), [(len(h), round(c, 4)) for h, c in zip(*res[dt])], flush=True)
    a, b = res[torch.float32], res[torch.float64]
    n = 0
    while n < min(len(a[0]), len(b[0])) and a[0][n] == b[0][n]:
        n += 1
    gaps = numpy.diff(numpy.array(b[1])) if len(b[1]) > 1 else numpy.array([])
    print(json.dumps(scales), "utt %d: identical ranked head %d of %d / %d; worst cost diff on the head %.2e; smallest gap between "
          "ranked costs (float64) %.2e" % (utt, n, len(a[0]), len(b[0]),
                                           max([abs(x_ - y_) for x_, y_ in zip(a[1][:n], b[1][:n])] or [0]), gaps.min() if len(gaps) else -1))
    print("first:", a[0][0] if a[0] else None)


if __name__ == "__main__":
    mode = sys.argv[1]
    scales = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
    n = int(sys.argv[3]) if len(sys.argv) > 3 else (4 if mode == "train" else 0)
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
    torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "4")))
    (train if mode == "train" else decode)(scales, n, seed, scale)
