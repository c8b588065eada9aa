"""GPU probe: lvsr_sgemm at (M, N) fixed and K swept, and square sizes: per-k-iteration slope vs per-tile fixed cost of the 128x128x32
MFMA kernel (csrc/gemm.hip).  python tools/probes/gemm_k_sweep.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "attention-lvcsr_amd")]
import torch
from lvsr_amd import native
lib = native.get()
lib.set_knobs([a for a in sys.argv[1:] if "=" in a]); sys.argv = [a for a in sys.argv if "=" not in a]
dev = torch.device("cuda:0")


def run(M, N, K, tA=False, tB=False, reps=20):
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    B = torch.randn((N, K) if tB else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(64 << 20, device=dev) if tA else None
    for _ in range(3):
        lib.sgemm(A, B, C, transA=tA, transB=tB, ws=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.sgemm(A, B, C, transA=tA, transB=tB, ws=ws)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print("tA=%d tB=%d M=%6d N=%5d K=%6d  tiles %5d (%.2f per CU)  %8.1f us  %6.1f TFLOP/s" % (tA, tB, M, N, K, tiles, tiles / 256.0, us, 2.0 * M * N * K / us / 1e6), flush=True)


for K in (128, 256, 512, 1024, 2048, 4096):
    run(12800, 512, K)
for K in (512, 2048):
    run(32768, 512, K)            # 1024 tiles = exactly 4 per CU
    run(16384, 512, K)            # 512 tiles = 2 per CU
    run(8192, 512, K)             # 256 tiles = 1 per CU
for n in (2048, 4096):
    run(n, n, n)
run(12800, 1536, 512); run(12800, 1536, 40); run(6400, 1536, 512)
run(12800, 512, 1536, tB=True); run(12800, 512, 512, tB=True); run(12800, 40, 1536, tB=True)
run(512, 1536, 12800, tA=True); run(512, 1536, 6400, tA=True); run(40, 1536, 12800, tA=True)
# sustained: the layer's projection for ~4 s back to back — what the clocks settle at under continuous fp32 MFMA load (the cool-chip
# figures above are 20 launches = 4 ms)
if "sustained" in sys.argv:
    M, N, K = 12800, 1536, 512
    A, Bm, C = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
    for chunk in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2500):
            lib.sgemm(A, Bm, C)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 2500
        print("sustained chunk %d (2500 launches): %7.1f us  %6.1f TFLOP/s" % (chunk, us, 2.0 * M * N * K / us / 1e6), flush=True)
