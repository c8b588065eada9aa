"""GPU probe: time one WSJ-shape BiGRU layer (T=800,B=16,H=256, I=512) fwd/bwd, graph vs eager."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import torch
from lvsr_amd import spec, synthetic, native
from lvsr_amd.params import ParameterStore, Workspace
from lvsr_amd.bricks import Encoder

dev = torch.device("cuda:0")
lib = native.get()
for (Hs, sub, T, B, F) in [([256], [1], 800, 16, 512), ([512], [1], 800, 8, 1024)]:
    cfg = dict(input_dim=F, num_phonemes=6, dims_bidir=Hs, subsample=sub, dim_dec=4, dim_matcher=7,
               attention_type="content", post_merge_dims=None, embed_outputs=True)
    params = synthetic.make_params(cfg, seed=3)
    store = ParameterStore(cfg, dev, params)
    x = torch.randn(T, B, F, device=dev)
    dy = torch.randn(T, B, 2 * Hs[-1], device=dev)
    stream = torch.cuda.Stream()
    for use_graph, persistent in ((True, False), (False, True), (True, True)):
        enc = Encoder(spec.Dims(cfg), store, lib, Workspace(dev), use_graph=use_graph, use_persistent=persistent)
        with torch.cuda.stream(stream):
            for it in range(3):
                t0 = time.time()
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                enc.apply(x, None)
                e1.record()
                enc.backward(dy)
                e2.record()
                torch.cuda.synchronize()
                print("H=%d B=%d T=%d graph=%d persistent=%d it=%d fwd %.3f ms (%.2f us/step) bwd %.3f ms (%.2f us/step) host %.1f ms" % (
                    Hs[0], B, T, use_graph, persistent, it, e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / T,
                    e1.elapsed_time(e2), e1.elapsed_time(e2) * 1e3 / T, (time.time() - t0) * 1e3), flush=True)
