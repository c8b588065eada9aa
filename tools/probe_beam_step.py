"""GPU probe: host time vs GPU time of one beam position (graph replay vs eager launches)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd")); sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy, torch
import bench_decode
rec, cfg = bench_decode.build("cuda:0", 16, "device")
x = numpy.random.RandomState(1).normal(size=(800, 40)).astype(numpy.float32)
gen = rec.generator
for mode in ("graph", "eager"):
    gen.use_graph = mode == "graph"
    with rec._on_stream():
        rec.compute_contexts(x[:, None, :])
        st = gen.beam_begin(16, rec.eos_label, 266, False, 1.0, 1e9, "optimistic_future_cost")
        for _ in range(4):
            gen.beam_step()
        torch.cuda.synchronize()
        N = 64
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(N):
            gen.beam_step()
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print("%s: host enqueue %.1f us/position, GPU span %.1f us/position, until drained %.1f us/position" % (
        mode, (t1 - t0) / N * 1e6, e0.elapsed_time(e1) * 1e3 / N, (t2 - t0) / N * 1e6), flush=True)
