import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/attention-lvcsr_amd']
import torch
from lvsr_amd import spec, synthetic, native
from lvsr_amd.params import ParameterStore, Workspace
from lvsr_amd.bricks import Encoder
dev = torch.device("cuda:0"); lib = native.get()
H, B, T = 256, 16, 800; F = 2 * H
cfg = dict(input_dim=F, num_phonemes=6, dims_bidir=[H], subsample=[1], dim_dec=4, dim_matcher=7, attention_type="content", post_merge_dims=None, embed_outputs=True)
store = ParameterStore(cfg, dev, synthetic.make_params(cfg, seed=3))
x = torch.randn(T, B, F, device=dev); stream = torch.cuda.Stream()
for name, flags in (("full (wave-private)", "0"), ("shared buffer+barrier", "32"), ("nosave", "1"), ("nowait", "8"), ("nodot", "16"), ("shared nodot", "48"), ("spread over XCDs", "2")):
    os.environ["LVSR_PERSIST_FLAGS"] = flags
    enc = Encoder(spec.Dims(cfg), store, lib, Workspace(dev), use_graph=True, use_persistent=True)
    best = 1e9
    with torch.cuda.stream(stream):
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); enc.apply(x, None); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    print("%-22s layer fwd %.3f ms  (%.2f us/step incl. 0.33 of projections)" % (name, best, best * 1e3 / T), flush=True)
