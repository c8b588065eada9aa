"""How well conditioned is the teacher-forced cost of the WSJ-base network under window_around_median(10, 100) with random weights?
float32 vs float64 ORACLE on the same batch: python tools/probes/wsj_train_conditioning.py SCALE B.  Result (seed 10, B = 4): the two
precisions agree to 1e-6 on the first labels and part ways after ~30-50 labels at scale 1.0 AND 2.0 (summed cost 1.1e-3 / 3.4e-3
relative, alignment argmax 92 % / 87 % equal): a window centre is a step function of the alignment.  No implementation pair can
be asserted label by label there; tests/test_gpu_properties.py compares the first labels tightly and the sum loosely."""
import sys, numpy, torch
import os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "attention-lvcsr_amd")]
from lvsr_amd import spec, synthetic
from oracle import lvsr_oracle as O
scale = float(sys.argv[1]); B = int(sys.argv[2])
cfg = spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100))
params = synthetic.make_params(cfg, seed=10, scale=scale)
batch = synthetic.make_batch(cfg, B, 800, 100, seed=1234)
res = {}
for dt in (torch.float32, torch.float64):
    orc = O.OracleRecognizer(cfg, params, dtype=dt)
    with torch.no_grad():
        out = orc.cost(batch["recordings"], batch["recordings_mask"], batch["labels"], batch["labels_mask"])
    res[dt] = (out["cost_matrix"].numpy().astype(numpy.float64), out["weights"].numpy())
a, b = res[torch.float32], res[torch.float64]
d = numpy.abs(a[0] - b[0])
print("scale", scale, "B", B, "cost sum rel", abs(a[0].sum() - b[0].sum()) / b[0].sum(), "max |dcost| per label:", d.max(axis=1)[[0, 5, 10, 20, 30, 50, 70, 99]])
print("argmax equal fraction", (a[1].argmax(2) == b[1].argmax(2)).mean(), "max weight mean", b[1].max(axis=2).mean())
