"""GPU probe: does the DATA decide the fp32-MFMA GEMM rate?  The same lvsr_sgemm launches on all-zero operands, on constant operands and on
N(0,1) operands (what every other measurement uses): matrix cores multiplying zeros toggle few wires — if the rate differs, the gap between
the instruction-rate probe (tools/probes/mfma_rate_probe.hip, constant operands) and the product kernel is the chip's power management,
not the kernel.  python tools/probes/gemm_data_probe.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "attention-lvcsr_amd")]
import torch
from lvsr_amd import native
lib = native.get()
lib.set_knobs([a for a in sys.argv[1:] if "=" in a])
dev = torch.device("cuda:0")


def run(M, N, K, fill, tA=False, tB=False, reps=40):
    mk = {"zeros": lambda *s: torch.zeros(*s, device=dev), "ones": lambda *s: torch.full(s, 1.0, device=dev),
          "randn": lambda *s: torch.randn(*s, device=dev), "small ints": lambda *s: torch.randint(0, 4, s, device=dev).float()}[fill]
    A = mk(*((K, M) if tA else (M, K)))
    B = mk(*((N, K) if tB else (K, N)))
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(64 << 20, device=dev) if tA else None
    for _ in range(5):
        lib.sgemm(A, B, C, transA=tA, transB=tB, ws=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.sgemm(A, B, C, transA=tA, transB=tB, ws=ws)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("tA=%d tB=%d M=%6d N=%5d K=%6d  %-10s %8.1f us  %6.1f TFLOP/s  (%.3f of 157.3)" % (tA, tB, M, N, K, fill, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / 157.3), flush=True)


for shape in ((16384, 512, 2048), (4096, 4096, 4096), (12800, 1536, 512)):
    for fill in ("zeros", "ones", "small ints", "randn", "zeros"):
        run(*shape, fill)
run(12800, 512, 1536, "zeros", tB=True); run(12800, 512, 1536, "randn", tB=True)
run(512, 1536, 12800, "zeros", tA=True); run(512, 1536, 12800, "randn", tA=True)
