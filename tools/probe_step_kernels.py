"""GPU probe: per-launch cost of each BiGRU step kernel in isolation (kernel_mask), hipGraph replay, H=256 B=16 T=800."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import torch
from lvsr_amd import spec, synthetic, native
from lvsr_amd.params import ParameterStore, Workspace
from lvsr_amd.bricks import Encoder

dev = torch.device("cuda:0")
lib = native.get()
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 800
B, F = 16, 64
cfg = dict(input_dim=F, num_phonemes=6, dims_bidir=[H], subsample=[1], dim_dec=4, dim_matcher=7,
           attention_type="content", post_merge_dims=None, embed_outputs=True)
store = ParameterStore(cfg, dev, synthetic.make_params(cfg, seed=3))
ws = Workspace(dev)
enc = Encoder(spec.Dims(cfg), store, lib, ws, use_graph=True, use_persistent=False)
pk = enc._packed(0)
p = store.p
nf, nb = enc._names(0, "forward"), enc._names(0, "backward")
z = lambda *s: torch.randn(*s, device=dev) * 0.1
bufs = dict(xg=z(T, B, 6 * H), y=z(T, B, 2 * H), u=torch.rand(T, B, 2 * H, device=dev), r=torch.rand(T, B, 2 * H, device=dev),
            c=z(T, B, 2 * H), rh=z(T, B, 2 * H), dy=z(T, B, 2 * H), dxg=z(T, B, 6 * H), dh_ws=z(12 * 16 * H))
stream = torch.cuda.Stream()
def timeit(fn):
    ts = []
    with torch.cuda.stream(stream):
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / T)
    return min(ts[1:])
for mask in (1, 2, 3):
    f = lambda: lib.run("lvsr_bigru_fwd", "lvsr_bigru_fwd_args", bufs["y"], True, xg=bufs["xg"], mask=None,
                        Whh_p=pk["Whh"], Whg_p=pk["Whg"], h0=[p[nf["h0"]], p[nb["h0"]]], y=bufs["y"], ysub=None, u=bufs["u"],
                        r=bufs["r"], c=bufs["c"], rh=bufs["rh"], sub=1, T=T, B=B, H=H, kernel_mask=mask)
    print("H=%d fwd kernel_mask=%d: %.2f us per step" % (H, mask, timeit(f)))
for mask in (1, 2, 3):
    f = lambda: lib.run("lvsr_bigru_bwd", "lvsr_bigru_bwd_args", bufs["dxg"], True, mask=None, y=bufs["y"], u=bufs["u"], r=bufs["r"],
                        c=bufs["c"], WhhT_p=pk["WhhT"], WhgT_p=pk["WhgT"], h0=[p[nf["h0"]], p[nb["h0"]]], dy=bufs["dy"],
                        dxg=bufs["dxg"], dh_ws=bufs["dh_ws"], dh0=[z(H), z(H)], sub=1, T=T, B=B, H=H, kernel_mask=mask)
    print("H=%d bwd kernel_mask=%d: %.2f us per step" % (H, mask, timeit(f)))

# hypothesis test: does a burst of dense GEMM work right before the graph slow the latency-bound chain (clock / cache state)?
X = torch.randn(12800, 512, device=dev); Wb = torch.randn(512, 1536, device=dev); Y = torch.empty(12800, 1536, device=dev)
def fwd_all():
    lib.run("lvsr_bigru_fwd", "lvsr_bigru_fwd_args", bufs["y"], True, xg=bufs["xg"], mask=None,
            Whh_p=pk["Whh"], Whg_p=pk["Whg"], h0=[p[nf["h0"]], p[nb["h0"]]], y=bufs["y"], ysub=None, u=bufs["u"],
            r=bufs["r"], c=bufs["c"], rh=bufs["rh"], sub=1, T=T, B=B, H=H, kernel_mask=3)
def with_gemm():
    for _ in range(4):
        lib.sgemm(X, Wb, Y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fwd_all(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / T
with torch.cuda.stream(stream):
    res = [with_gemm() for _ in range(4)]
print("H=%d fwd both kernels right after 4 big GEMMs: %s us per step" % (H, ", ".join("%.2f" % r for r in res)))
maskt = torch.ones(T, B, device=dev)
def fwd_mask():
    lib.run("lvsr_bigru_fwd", "lvsr_bigru_fwd_args", bufs["y"], True, xg=bufs["xg"], mask=maskt,
            Whh_p=pk["Whh"], Whg_p=pk["Whg"], h0=[p[nf["h0"]], p[nb["h0"]]], y=bufs["y"], ysub=None, u=bufs["u"],
            r=bufs["r"], c=bufs["c"], rh=bufs["rh"], sub=1, T=T, B=B, H=H, kernel_mask=3)
print("H=%d fwd both kernels with an all-ones mask: %.2f us per step" % (H, timeit(fwd_mask)))
bufs["xg"].normal_(0, 2.0)
print("H=%d fwd both kernels, xg ~ N(0,2) (mixed tanh/exp ranges): %.2f us per step" % (H, timeit(fwd_all)))

# back-to-back like the real encoder forward: [4 GEMMs, graph] x 4 without host syncs in between
def four_layers():
    evs = []
    for layer in range(4):
        for _ in range(4):
            lib.sgemm(X, Wb, Y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fwd_all(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 / T for a, b in evs]
with torch.cuda.stream(stream):
    four_layers()
    print("H=%d four back-to-back [GEMMs + fwd graph]: us per step per layer:" % H, ["%.2f" % v for v in four_layers()])

# hypothesis: in the real step every layer has its own ~200 MB of activations, so the working set (~1 GB) never stays in the
# 256 MB Infinity Cache; this probe re-used one set.  Rotate over 5 sets.
sets = []
for k in range(5):
    sets.append(dict(xg=z(T, B, 6 * H), y=z(T, B, 2 * H), u=torch.rand(T, B, 2 * H, device=dev), r=torch.rand(T, B, 2 * H, device=dev),
                     c=z(T, B, 2 * H), rh=z(T, B, 2 * H)))
def fwd_set(bs):
    lib.run("lvsr_bigru_fwd", "lvsr_bigru_fwd_args", bs["y"], True, xg=bs["xg"], mask=None,
            Whh_p=pk["Whh"], Whg_p=pk["Whg"], h0=[p[nf["h0"]], p[nb["h0"]]], y=bs["y"], ysub=None, u=bs["u"],
            r=bs["r"], c=bs["c"], rh=bs["rh"], sub=1, T=T, B=B, H=H, kernel_mask=3)
def rotate():
    evs = []
    for bs in sets:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fwd_set(bs); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 / T for a, b in evs]
with torch.cuda.stream(stream):
    rotate()
    print("H=%d fwd over 5 rotating buffer sets (1 GB working set): us per step:" % H, ["%.2f" % v for v in rotate()])
