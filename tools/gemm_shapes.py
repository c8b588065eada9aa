"""GPU probe: record every lvsr_sgemm shape one WSJ-base training step issues, then time each distinct shape alone
(HIP events, 30 launches) -> TFLOP/s per shape and the step's total GEMM time if each ran at that speed."""
import collections, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import torch
from lvsr_amd import spec, synthetic, native
from lvsr_amd.bricks.recognizer import SpeechRecognizer

dev = torch.device("cuda:0")
factory, WB, WT, WL = spec.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "wsj_base"]
cfg = factory()
rec = SpeechRecognizer(device=dev, params=synthetic.make_params(cfg, seed=1), net_config=cfg)
batch = synthetic.make_batch(cfg, WB, WT, WL, seed=2, ragged=False)
lib = rec.lib
shapes = collections.Counter()
orig = lib.sgemm
def spy(A, B, C, transA=False, transB=False, alpha=1.0, beta=0.0, bias=None, ws=None, M=None, N=None, K=None, **kw):
    m = M if M is not None else (A.shape[1] if transA else A.shape[0])
    k = K if K is not None else (A.shape[0] if transA else A.shape[1])
    n = N if N is not None else (B.shape[0] if transB else B.shape[1])
    shapes[(int(transA), int(transB), m, n, k, ws is not None)] += 1
    return orig(A, B, C, transA=transA, transB=transB, alpha=alpha, beta=beta, bias=bias, ws=ws, M=M, N=N, K=K, **kw)
lib.sgemm = spy
rec.cost_and_gradients(batch)
torch.cuda.synchronize()
lib.sgemm = orig
ws = torch.empty(64 << 20, device=dev)
tot_t = tot_f = 0.0
print("tA tB      M      N      K  ws  calls   us/call  TFLOP/s")
for (ta, tb, m, n, k, has_ws), cnt in sorted(shapes.items(), key=lambda kv: -kv[1] * kv[0][2] * kv[0][3] * kv[0][4]):
    A = torch.randn((k, m) if ta else (m, k), device=dev)
    B = torch.randn((n, k) if tb else (k, n), device=dev)
    C = torch.empty(m, n, device=dev)
    for _ in range(3):
        orig(A, B, C, transA=bool(ta), transB=bool(tb), ws=ws if has_ws else None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        orig(A, B, C, transA=bool(ta), transB=bool(tb), ws=ws if has_ws else None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    fl = 2.0 * m * n * k
    tot_t += us * cnt; tot_f += fl * cnt
    print("%2d %2d %6d %6d %6d %3d %6d %9.1f %8.1f" % (ta, tb, m, n, k, has_ws, cnt, us, fl / us * 1e-6))
print("step total: %.2f ms of GEMM, %.1f GFLOP, %.1f TFLOP/s average (fp32 MFMA peak 157.3)" % (tot_t * 1e-3, tot_f * 1e-9, tot_f / tot_t * 1e-6))
