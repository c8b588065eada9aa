"""GPU probe: per-k-tile slope and fixed cost of the 128 x 128 tile kernel on a one-round grid (16384 x 512: 512 tiles = 2 per CU), zeros."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "attention-lvcsr_amd")]
import torch
from lvsr_amd import native
lib = native.get()
lib.set_knobs(["gemm_mid_tiles=1"] + [a for a in sys.argv[1:] if "=" in a])
dev = torch.device("cuda:0")
prev = None
for tA, tB in ((False, False), (False, True), (True, False)):
    for K in (512, 1024, 2048, 4096, 8192):
        M, N = 16384, 512
        A = torch.zeros((K, M) if tA else (M, K), device=dev); B = torch.zeros((N, K) if tB else (K, N), device=dev); C = torch.empty(M, N, device=dev)
        for _ in range(5):
            lib.sgemm(A, B, C, transA=tA, transB=tB)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.sgemm(A, B, C, transA=tA, transB=tB)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        slope = "" if prev is None or prev[0] >= K else "  slope %.3f us per k-tile of 32" % ((us - prev[1]) / ((K - prev[0]) / 32))
        print("tA=%d tB=%d 16384 x 512 x %5d: %8.1f us  %6.1f TFLOP/s%s" % (tA, tB, K, us, 2.0 * M * N * K / us / 1e6, slope), flush=True)
        prev = (K, us)
    prev = None
