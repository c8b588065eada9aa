"""Small eager (no hipGraph) training step at WSJ-base layer shapes but few time steps, for rocprofv3 --pmc passes
(counter collection serialises every dispatch; the full bench with its 12k graph-launched kernels does not finish)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import torch
from lvsr_amd import spec, synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
from lvsr_amd.training import Trainer
import bench
cfg = spec.wsj_base()
T, L, B = int(os.environ.get("PMC_T", "64")), int(os.environ.get("PMC_L", "8")), 16
dev = torch.device("cuda:0")
rec = SpeechRecognizer(device=dev, params=synthetic.make_params(cfg, seed=10), net_config=cfg, use_graph=False)
tr = Trainer(rec, distributed=False, **bench.TRAIN_CONF)
batch = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(cfg, B, T, L, seed=1234).items()}
for _ in range(2):
    cm = tr.train_step(batch)
torch.cuda.synchronize()
print("pmc probe done, cost", float(cm.sum()))
