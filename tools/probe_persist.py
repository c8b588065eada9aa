"""GPU probe: one WSJ-shape BiGRU layer (T=800), step kernels under hipGraph vs the persistent cluster kernels
(csrc/encoder_persist.hip) for several utterances-per-cluster / experiment-flag settings; us per time step.
    python tools/probe_persist.py [H B T] ...
"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import torch
from lvsr_amd import spec, synthetic, native
from lvsr_amd.params import ParameterStore, Workspace
from lvsr_amd.bricks import Encoder

dev = torch.device("cuda:0")
# The ablation variants below (flags 1, 8, 16, 128: wrong results, timing only) exist only in the probe build of the library
# (python attention-lvcsr_amd/csrc/build.py --probes -> tools/probes/liblvsr_hip_probes.so); the product library refuses them.
PROBES = os.path.join(REPO, "tools", "probes", "liblvsr_hip_probes.so")
lib = native.Lib(PROBES) if os.path.exists(PROBES) else native.get()
ABLATIONS = lib is not native._default
BASE_KNOBS = [a for a in sys.argv[1:] if "=" in a]          # e.g. python tools/probe_persist.py max_cluster_wgs=256 512 8 1500
sys.argv = [a for a in sys.argv if "=" not in a]
lib.set_knobs(BASE_KNOBS)
shapes = [(256, 16, 800), (512, 8, 800), (128, 2, 200), (250, 16, 800)]
if len(sys.argv) >= 4:
    shapes = [tuple(int(v) for v in sys.argv[i:i + 3]) for i in range(1, len(sys.argv) - 2, 3)]
# name, persistent?, knob persist_rows, knob persist_flags (persist.h: 1 no saves, 2 clusters spread over the XCDs, 4 write-through
# stores even inside an XCD = the round-2 hand-off, 8 no waiting, 16 no contractions, 64 clusters of 4 instead of 8 work-groups)
variants = [("steps+graph", False, None, None), ("persist", True, "0", "0"), ("persist, write-through stores", True, "0", "4"),
            ("persist, no waiting", True, "0", "8"), ("persist, no dots", True, "0", "16"), ("persist rows=2", True, "2", "0"),
            ("persist rows=2, write-through", True, "2", "4"), ("persist rows=4", True, "4", "0"), ("persist nosave", True, "0", "1"),
            ("persist spread over XCDs", True, "0", "2"), ("persist 256 threads", True, "0", "0", "256"),
            ("persist 256 threads, write-through", True, "0", "4", "256"), ("persist, owners fetch (no loader waves)", True, "0", "256"), ("persist, one unit per lane group (H > 256)", True, "0", "2048"), ("persist, no prefetch", True, "0", "128"), ("persist, no prefetch, no saves", True, "0", "129"),
            ("persist, no operand fetches", True, "0", "128"), ("persist, no operand fetches, no saves", True, "0", "129"),
            ("persist, no prefetch, no saves, no dots", True, "0", "145"), ("persist, clusters of 4", True, "0", "64"),
            ("persist, clusters of 4, no dots", True, "0", "80"), ("persist, clusters of 4, write-through", True, "0", "68")]
for (H, B, T) in shapes:
    F = 2 * H
    cfg = dict(input_dim=F, num_phonemes=6, dims_bidir=[H], subsample=[1], dim_dec=4, dim_matcher=7,
               attention_type="content", post_merge_dims=None, embed_outputs=True)
    params = synthetic.make_params(cfg, seed=3)
    store = ParameterStore(cfg, dev, params)
    x = torch.randn(T, B, F, device=dev)
    dy = torch.randn(T, B, 2 * H, device=dev)
    stream = torch.cuda.Stream()
    ref = None
    for name, persistent, rows, flags, *more in variants:
        if flags and int(flags) & (1 | 8 | 16 | 128) and not ABLATIONS:
            continue
        for k, v in (("persist_rows", rows), ("persist_flags", flags), ("persist_threads", more[0] if more else None)):
            lib.set_knob(k, int(v or 0))
        enc = Encoder(spec.Dims(cfg), store, lib, Workspace(dev), use_graph=True, use_persistent=persistent)
        best = [1e9, 1e9]
        with torch.cuda.stream(stream):
            for it in range(4):
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                # the input projections (GEMMs) are part of apply(): time the recurrence alone through a second apply of
                # the layer call is not possible from here, so report the whole layer and the GEMM share separately below
                e0.record()
                y, _ = enc.apply(x, None)
                e1.record()
                enc.backward(dy)
                e2.record()
                torch.cuda.synchronize()
                best = [min(best[0], e0.elapsed_time(e1)), min(best[1], e1.elapsed_time(e2))]
        if persistent:
            try:
                enc.check_persistent()
            except Exception as exc:          # a cluster gave up (e.g. 256 work-groups not co-resident): report and go on
                print("H=%d B=%d T=%d %-26s FAILED: %s" % (H, B, T, name, exc), flush=True)
                continue
        gx = store.g["/recognizer/encoder/bidir0/forward/fork/fork_inputs.W"].clone()
        err = ""
        if ref is None:
            ref = (y.clone(), gx)
        elif flags in ("0", "2", "4", "64", "68", "256", "2048"):
            err = "  max|dy| %.2e  max|dgrad| %.2e (rel %.1e)" % (float((y - ref[0]).abs().max()), float((gx - ref[1]).abs().max()),
                                                              float((gx - ref[1]).abs().max() / ref[1].abs().max()))
        print("H=%d B=%d T=%d %-26s layer fwd %.3f ms (%.2f us/step)  bwd %.3f ms (%.2f us/step)%s" % (
            H, B, T, name, best[0], best[0] * 1e3 / T, best[1], best[1] * 1e3 / T, err), flush=True)
    # GEMM share of the layer (input projections forward; input/weight gradient GEMMs backward), for subtraction
    with torch.cuda.stream(stream):
        A = torch.randn(T * B, F, device=dev); W = torch.randn(F, 3 * H, device=dev); C = torch.empty(T * B, 3 * H, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lib.sgemm(A, W, C)
        e0.record()
        for _ in range(4):
            lib.sgemm(A, W, C)
        e1.record(); torch.cuda.synchronize()
        print("   (forward input projections of the layer ~ %.3f ms)" % (e0.elapsed_time(e1) / 2), flush=True)
