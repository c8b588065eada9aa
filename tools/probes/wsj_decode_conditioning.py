"""How well conditioned is a full-size WSJ decode (configs[4]) at a given parameter scale?  Runs the float32 and the float64
torch ORACLE beam search (beam 16, FST LM, window_around_median(10, 100), exp/wsj/decode.sh settings) on the same utterance and
prints where the ranked hypotheses part.  Used to choose the scale of tests/golden/wsj_decode_full.npz (gen_golden.py).
    python tools/probes/wsj_decode_conditioning.py SCALE [UTT] [T]
"""
import math
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


def lm_arcs(V, fst_seed):
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


if __name__ == "__main__":
    scale = float(sys.argv[1])
    utt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 800
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    cfg = spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100))
    cfg["max_decoded_length_scale"] = 3.0
    V = cfg["num_phonemes"]
    params = synthetic.make_params(cfg, seed=seed, scale=scale)
    arcs = lm_arcs(V, 9)
    lm = dict(dense=LO.DenseFST(arcs, 0, V), remap={c: c + 1 for c in range(V)}, no_transition_cost=20.0, weight=0.5)
    x = numpy.random.RandomState(100 + utt).normal(size=(T, cfg["input_dim"])).astype("float32")
    res = {}
    for dt in (torch.float32, torch.float64):
        t0 = time.time()
        orc = O.OracleRecognizer(cfg, params, dtype=dt)
        try:
            res[dt] = orc.beam_search(x, 16, char_discount=1.0, round_to_inf=1e9, stop_on="optimistic_future_cost", lm=lm)
        except LookupError as e:
            res[dt] = ([], [])
        print(dt, "%.0fs" % (time.time() - t0), [(len(h), round(c, 3)) for h, c in zip(*res[dt])][:16], flush=True)
    a, b = res[torch.float32], res[torch.float64]
    n = 0
    while n < min(len(a[0]), len(b[0])) and a[0][n] == b[0][n]:
        n += 1
    print("scale %g utt %d: identical ranked head: %d of %d / %d" % (scale, utt, n, len(a[0]), len(b[0])))
    print("first:", a[0][0] if a[0] else None)
