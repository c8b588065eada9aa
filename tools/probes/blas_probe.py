"""GPU probe: library fp32 GEMM (torch.mm -> rocBLAS / hipBLASLt) vs lvsr_sgemm on the big projection shapes of a WSJ-base step."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import torch
from lvsr_amd import native
lib = native.get()
torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda:0"
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, K, ta, tb) in [(12800, 1536, 512, 0, 0), (12800, 512, 1536, 0, 1), (512, 1536, 12800, 1, 0), (12800, 512, 512, 0, 0), (6400, 1536, 512, 0, 0), (3200, 768, 512, 0, 0)]:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev); C = torch.empty(M, N, device=dev)
    ws = torch.empty(1 << 22, device=dev)
    mine = t(lambda: lib.sgemm(A, B, C, transA=bool(ta), transB=bool(tb), ws=ws))
    Ao, Bo = (A.t() if ta else A), (B.t() if tb else B)
    libt = t(lambda: torch.mm(Ao, Bo, out=C))
    fl = 2.0 * M * N * K
    print("M=%5d N=%4d K=%5d tA=%d tB=%d: lvsr_sgemm %6.1f us (%5.1f TF)   torch.mm %6.1f us (%5.1f TF)" % (M, N, K, ta, tb, mine, fl / mine / 1e6, libt, fl / libt / 1e6), flush=True)
