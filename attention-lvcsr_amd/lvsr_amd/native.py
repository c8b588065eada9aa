"""ctypes binding of the C-ABI library (include/lvsr_hip.h) — the only way the Python host side reaches
the HIP kernels.  No torch types cross the boundary: tensors are passed as raw device pointers + sizes,
the stream as a `hipStream_t` handle (torch's current stream on the tensor's device).

The library is mandatory: if `liblvsr_hip.so` is missing or a call fails this raises; there is no
PyTorch/CPU fallback anywhere in the product path.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "liblvsr_hip.so")

c_void_p, c_int, c_float, c_ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong
P = c_void_p

# name -> (restype, argtypes).  Must list every symbol declared in include/lvsr_hip.h
# (tests/test_abi.py parses the header and checks both directions).
SIGNATURES = {
    "lvsr_last_error": (ctypes.c_char_p, []),
    "lvsr_abi_version": (c_int, []),
    "lvsr_sgemm": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_float, P, c_int, P, c_int, c_float, P, c_int, P, P, c_ll]),
    "lvsr_colsum": (c_int, [P, P, c_int, c_int, c_int, P, c_float]),
    "lvsr_transpose": (c_int, [P, P, c_int, c_int, P]),
    "lvsr_graph_clear": (None, []),
    "lvsr_graph_count": (c_int, []),
    "lvsr_bigru_fwd": (c_int, [P] + [P] * 8 + [P, P, c_int] + [P] * 4 + [c_int] * 4),
    "lvsr_bigru_bwd": (c_int, [P] + [P] * 5 + [P] * 6 + [P, c_int, P, P, P, P] + [c_int] * 4),
}


class NativeError(RuntimeError):
    pass


def ptr(t):
    """Raw pointer of a tensor (or None)."""
    if t is None:
        return None
    if not t.is_contiguous() and t.dim() > 0 and t.numel() > 0:
        # strided views are passed with explicit leading dimensions by the callers; they must
        # at least be dense in the last dimension.
        assert t.stride(-1) == 1, "last dimension must be contiguous"
    return ctypes.c_void_p(t.data_ptr())


class Lib(object):
    def __init__(self, path=DEFAULT_LIB, signatures=None):
        if not os.path.exists(path):
            raise NativeError(
                "HIP extension %s not found: build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). There is no fallback path." % path)
        self.path = path
        self._dll = ctypes.CDLL(path)
        self.is_emulator = os.path.basename(path) != os.path.basename(DEFAULT_LIB)
        for name, (res, args) in (signatures or SIGNATURES).items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                raise NativeError("symbol %s missing from %s" % (name, path))
            fn.restype = res
            fn.argtypes = args
            setattr(self, "_" + name, fn)

    def stream_for(self, t):
        if t.is_cuda:
            return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
        if not self.is_emulator:
            raise NativeError("CPU tensor passed to the HIP library; the hot path runs on cuda devices only")
        return ctypes.c_void_p(0)

    def call(self, name, *args):
        rc = getattr(self, "_" + name)(*args)
        if rc != 0:
            raise NativeError("%s failed (%d): %s" % (name, rc, self._lvsr_last_error().decode()))

    def last_error(self):
        return self._lvsr_last_error().decode()

    # ---- thin typed wrappers -----------------------------------------------------------------
    def sgemm(self, A, B, C, transA=False, transB=False, alpha=1.0, beta=0.0, bias=None, ws=None,
              M=None, N=None, K=None, lda=None, ldb=None, ldc=None):
        """C = alpha*op(A)@op(B) + beta*C + bias.  A,B,C are 2-D (possibly strided-row) fp32 tensors."""
        if M is None:
            M = A.shape[1] if transA else A.shape[0]
        if K is None:
            K = A.shape[0] if transA else A.shape[1]
        if N is None:
            N = B.shape[0] if transB else B.shape[1]
        lda = A.stride(0) if lda is None else lda
        ldb = B.stride(0) if ldb is None else ldb
        ldc = C.stride(0) if ldc is None else ldc
        self.call("lvsr_sgemm", self.stream_for(C), int(transA), int(transB), M, N, K, alpha, ptr(A), lda, ptr(B), ldb,
                  beta, ptr(C), ldc, ptr(bias), ptr(ws), (ws.numel() * 4 if ws is not None else 0))

    def colsum(self, X, out, beta=0.0, M=None, N=None, ldx=None):
        M = X.shape[0] if M is None else M
        N = X.shape[1] if N is None else N
        self.call("lvsr_colsum", self.stream_for(out), ptr(X), M, N, X.stride(0) if ldx is None else ldx, ptr(out), beta)

    def transpose(self, x, out):
        self.call("lvsr_transpose", self.stream_for(out), ptr(x), x.shape[0], x.shape[1], ptr(out))


    def bigru_fwd(self, xg, mask, Wf, Wb, y, ysub, sub, u, r, c, rh, T, B, H, use_graph):
        """Wf/Wb = (state_to_state, state_to_gates, initial_state) of the forward / backward direction."""
        self.call("lvsr_bigru_fwd", self.stream_for(y), ptr(xg), ptr(mask), ptr(Wf[0]), ptr(Wf[1]), ptr(Wf[2]),
                  ptr(Wb[0]), ptr(Wb[1]), ptr(Wb[2]), ptr(y), ptr(ysub), sub, ptr(u), ptr(r), ptr(c), ptr(rh),
                  T, B, H, int(use_graph))

    def bigru_bwd(self, mask, y, u, r, c, WTf, WTb, dy, sub, dxg, dh_ws, dh0_f, dh0_b, T, B, H, use_graph):
        """WTf/WTb = (state_to_state^T, state_to_gates^T, initial_state)."""
        self.call("lvsr_bigru_bwd", self.stream_for(dxg), ptr(mask), ptr(y), ptr(u), ptr(r), ptr(c), ptr(WTf[0]),
                  ptr(WTf[1]), ptr(WTf[2]), ptr(WTb[0]), ptr(WTb[1]), ptr(WTb[2]), ptr(dy), sub, ptr(dxg), ptr(dh_ws),
                  ptr(dh0_f), ptr(dh0_b), T, B, H, int(use_graph))


_default = None


def get():
    """The product library (gfx950).  Raises if it is not built."""
    global _default
    if _default is None:
        _default = Lib(DEFAULT_LIB)
    return _default
