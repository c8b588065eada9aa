"""The batched front-end kernels alone, for rocprofv3 --pmc passes (tools/gpu_session.sh fbpmc): 512 utterances x 8 s of PCM, 20 launches
of lvsr_fbank_batch + lvsr_add_deltas_cmvn_batch (bench.py's `fbank` leg)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "attention-lvcsr_amd")):
    sys.path.insert(0, p)
import torch

import bench

out = bench.fbank_leg(torch.device("cuda:0"))
print({k: (v.get("launch_us"), v.get("achieved"), v.get("frac")) for k, v in out.items() if isinstance(v, dict)})
