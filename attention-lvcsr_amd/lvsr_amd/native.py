"""ctypes binding of the C-ABI library (include/lvsr_hip.h) — the only way the Python host side reaches
the HIP kernels.  No torch types cross the boundary: tensors are passed as raw device pointers + sizes,
the stream as a `hipStream_t` handle.

The binding is generated from the header itself (structs and prototypes are parsed from
include/lvsr_hip.h), so the header is the single source of truth for the ABI.

The library is mandatory: if `liblvsr_hip.so` is missing or a call fails this raises; there is no
PyTorch/CPU fallback anywhere in the product path.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "liblvsr_hip.so")
HEADER = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "lvsr_hip.h")

# Measured on MI355X / ROCm 7.2 (WSJ-base step): when the host keeps submitting work while a 1600-node time-loop graph
# replays, the graph's kernels run ~25 % slower (6.2 -> 8.0 us per encoder step); blocking the host until the graph has
# drained gives 57.3 ms per step instead of 66.4 (sync BEFORE the launch: 58.6; a dedicated graph stream: 71 — both removed).  Default on.
# `Lib.sync_after_graph` (constructor argument, default True).
#
# Whole-step graph regions (lvsr_region_begin/end): one hipGraph launch per training step.  `Lib.step_graph = False` (constructor
# argument `step_graph`) keeps the per-layer time-loop graphs with eager launches between them.

_SCALARS = {"int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double,
            "long long": ctypes.c_longlong, "char": ctypes.c_char}


class NativeError(RuntimeError):
    pass


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    return src


def _ctype(base, nptr, structs):
    base = base.replace("const", " ").replace("struct", " ")
    base = " ".join(base.split())
    if nptr:
        if base == "char" and nptr == 1:
            return ctypes.c_char_p
        return ctypes.c_void_p
    if base == "void":
        return None
    if base in _SCALARS:
        return _SCALARS[base]
    if base in structs:
        return structs[base]
    raise NativeError("unsupported C type %r in %s" % (base, HEADER))


def parse_header(path=HEADER):
    """-> (structs: name -> ctypes.Structure subclass, functions: name -> (restype, [argtypes], [argnames]))."""
    src = _strip_comments(open(path).read())
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            dm = re.match(r"((?:const\s+)?(?:unsigned\s+)?(?:long\s+long|\w+))\s*(.*)$", decl, flags=re.S)
            base, rest = dm.group(1), dm.group(2)
            for d in rest.split(","):
                d = d.strip()
                nptr = d.count("*")
                d = d.replace("*", "").strip()
                am = re.match(r"(\w+)\s*\[(\d+)\]$", d)
                ct = _ctype(base, nptr, structs)
                if am:
                    fields.append((am.group(1), ct * int(am.group(2))))
                else:
                    fields.append((d, ct))
        structs[name] = type(name, (ctypes.Structure,), {"_fields_": fields})
    src_nostruct = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    src_nostruct = re.sub(r'extern\s+"C"\s*\{', " ", src_nostruct)
    functions = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(lvsr_\w+)\s*\(([^;{}]*?)\)\s*;", src_nostruct, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        restype = _ctype(ret.replace("*", ""), ret.count("*"), structs)
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                nptr = a.count("*")
                a = a.replace("*", " ")
                toks = a.split()
                argnames.append(toks[-1])
                base = " ".join(toks[:-1])
                if nptr and base.replace("const", "").strip() in structs:
                    argtypes.append(ctypes.POINTER(structs[base.replace("const", "").strip()]))
                else:
                    argtypes.append(_ctype(base, nptr, structs))
        functions[name] = (restype, argtypes, argnames)
    return structs, functions


def ptr(t):
    """Raw address of a tensor (0 for None)."""
    if t is None:
        return None
    if isinstance(t, int):
        return ctypes.c_void_p(t)
    if t.dim() > 0 and t.numel() > 0:
        assert t.stride(-1) == 1, "last dimension must be contiguous"
    return ctypes.c_void_p(t.data_ptr())


def _addr(v):
    if v is None:
        return None
    if isinstance(v, int):
        return v
    if v.dim() > 0 and v.numel() > 0:
        assert v.stride(-1) == 1, "last dimension must be contiguous"
    return v.data_ptr()


class Lib(object):
    def __init__(self, path=DEFAULT_LIB, sync_after_graph=True, step_graph=True):
        self.sync_after_graph = bool(sync_after_graph)
        self.step_graph = bool(step_graph)
        if not os.path.exists(path):
            raise NativeError(
                "HIP extension %s not found: build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). There is no fallback path." % path)
        self.path = path
        self._dll = ctypes.CDLL(path)
        self.is_emulator = hasattr(self._dll, "hipemu_set_concurrent")      # tests/hipemu build of the same sources (CPU fibers)
        self._gstream = None
        self.capturing = False          # inside a Region capture: no host synchronisation, no nested graphs
        self.structs, self.functions = parse_header()
        self.knobs = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+LVSR_KNOB_(\w+)\s+(\d+)", open(HEADER).read())
                      if m.group(1) != "COUNT"}
        for name, (res, args, _) in self.functions.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                raise NativeError("symbol %s (declared in include/lvsr_hip.h) missing from %s" % (name, path))
            fn.restype = res
            fn.argtypes = args
            setattr(self, "_" + name, fn)

    # ---- plumbing -----------------------------------------------------------------------------
    def emulates_concurrency(self):
        """Test emulator only: are work-groups of a launch run concurrently (tests/hipemu, hipemu_set_concurrent)?"""
        if not self.is_emulator or not hasattr(self._dll, "hipemu_get_concurrent"):
            return False
        return bool(self._dll.hipemu_get_concurrent())

    def stream_for(self, t):
        if t.is_cuda:
            return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
        if not self.is_emulator:
            raise NativeError("CPU tensor passed to the HIP library; the hot path runs on cuda devices only")
        return ctypes.c_void_p(0)

    def make(self, struct_name, **kw):
        """Build an argument block; tensors become raw addresses, lists fill pointer arrays."""
        cls = self.structs[struct_name]
        s = cls()
        known = dict(cls._fields_)
        for k, v in kw.items():
            if k not in known:
                raise NativeError("%s has no field %s" % (struct_name, k))
            ft = known[k]
            if isinstance(v, (list, tuple)):
                arr = getattr(s, k)
                for i, e in enumerate(v):
                    arr[i] = _addr(e) if ft._type_ is ctypes.c_void_p else e
            elif ft is ctypes.c_void_p:
                setattr(s, k, _addr(v))
            else:
                setattr(s, k, v)
        return s

    def call(self, name, *args):
        rc = getattr(self, "_" + name)(*args)
        if rc != 0:
            raise NativeError("%s failed (%d): %s" % (name, rc, self._lvsr_last_error().decode()))

    def last_error(self):
        return self._lvsr_last_error().decode()

    # ---- tuning knobs (include/lvsr_hip.h LVSR_KNOB_*): kernel-variant switches for probes and A/B measurements ------------
    def set_knob(self, name, value):
        """name: 'persist_rows', 'persist_flags', 'persist_threads', 'phase_clock', 'max_cluster_wgs' (or the integer id)."""
        self.call("lvsr_set_knob", name if isinstance(name, int) else self.knobs[name.upper()], int(value))

    def get_knob(self, name):
        return int(self._lvsr_get_knob(name if isinstance(name, int) else self.knobs[name.upper()]))

    def set_knobs(self, settings):
        """settings: {knob name: int} or an iterable of "name=value" strings (the --knob arguments of bench.py and tools/);
        knobs that are not named are reset to 0."""
        if not isinstance(settings, dict):
            settings = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in settings)
        unknown = [k for k in settings if k.upper() not in self.knobs]
        if unknown:
            raise NativeError("unknown knob(s) %s (include/lvsr_hip.h LVSR_KNOB_*: %s)" % (unknown, sorted(k.lower() for k in self.knobs)))
        low = {k.upper(): int(v) for k, v in settings.items()}
        for name in self.knobs:
            self.set_knob(name, low.get(name, 0))

    # ---- thin typed wrappers -----------------------------------------------------------------
    # ---- grouped weight-gradient products ---------------------------------------------------------
    def begin_group(self):
        """Until flush_group(): sgemm(..., transA=True, group=True) calls are collected instead of launched."""
        self._group, self._group_after = [], []
        self._group_colsums = []

    def flush_group(self, ws):
        """One lvsr_sgemm_tn_grouped launch for everything collected since begin_group() (then the deferred follow-ups, in
        order).  The operands must still hold what they held when the products were requested."""
        jobs, after = getattr(self, "_group", None), getattr(self, "_group_after", None)
        self._group = self._group_after = None
        if jobs:
            cls = self.structs["lvsr_gemm_desc"]
            arr = (cls * len(jobs))()
            for d, (A, B, C, beta) in zip(arr, jobs):
                d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
                d.M, d.N, d.K = A.shape[1], B.shape[1], A.shape[0]
                d.lda, d.ldb, d.ldc, d.beta = A.stride(0), B.stride(0), C.stride(0), beta
            self.call("lvsr_sgemm_tn_grouped", self.stream_for(jobs[0][2]), arr, len(jobs), ptr(ws), ws.numel() * 4)
        after = list(after or ())
        # the follow-ups (rank-B updates with beta = 1 onto outputs of the grouped launch: K = batch rows, eight of them per step at 9.6 us
        # a launch) go out as ONE more grouped launch when they are plain transposed-A products onto distinct outputs (round 6)
        plain = [kw for kw in after if kw["transA"] and not kw["transB"] and kw["alpha"] == 1.0 and kw["bias"] is None
                 and kw["A"].stride(-1) == 1 and kw["B"].stride(-1) == 1 and kw["C"].stride(-1) == 1]
        if len(plain) == len(after) and len(after) > 1 and len({kw["C"].data_ptr() for kw in after}) == len(after):
            cls = self.structs["lvsr_gemm_desc"]
            arr = (cls * len(after))()
            for d, kw in zip(arr, after):
                A, B, C = kw["A"], kw["B"], kw["C"]
                d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
                d.M = kw["M"] if kw["M"] is not None else A.shape[1]
                d.K = kw["K"] if kw["K"] is not None else A.shape[0]
                d.N = kw["N"] if kw["N"] is not None else B.shape[1]
                d.lda = kw["lda"] if kw["lda"] is not None else A.stride(0)
                d.ldb = kw["ldb"] if kw["ldb"] is not None else B.stride(0)
                d.ldc = kw["ldc"] if kw["ldc"] is not None else C.stride(0)
                d.beta = float(kw["beta"])
            self.call("lvsr_sgemm_tn_grouped", self.stream_for(after[0]["C"]), arr, len(after), None, 0)
        else:
            for kw in after:
                self.sgemm(**kw)
        self._flush_colsums(ws)

    def sgemm(self, A, B, C, transA=False, transB=False, alpha=1.0, beta=0.0, bias=None, ws=None,
              M=None, N=None, K=None, lda=None, ldb=None, ldc=None, group=False):
        """C = alpha*op(A)@op(B) + beta*C + bias.  A,B,C are 2-D (possibly strided-row) fp32 tensors.
        group=True (weight gradients, transA only): join the pending grouped launch if one is open (begin_group)."""
        if group and getattr(self, "_group", None) is not None:
            plain = transA and not transB and alpha == 1.0 and bias is None and M is None and K is None and lda is None
            if plain and A.stride(1) == 1 and B.stride(1) == 1 and C.stride(1) == 1:
                self._group.append((A, B, C, float(beta)))
            else:       # depends on a collected product (e.g. a rank-B update with beta = 1): after the grouped launch, in order
                self._group_after.append(dict(A=A, B=B, C=C, transA=transA, transB=transB, alpha=alpha, beta=beta, bias=bias, ws=ws,
                                              M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc))
            return
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

    def colsum(self, X, out, beta=0.0, M=None, N=None, ldx=None, ws=None):
        """out[n] = beta * out[n] + sum_m X[m, n].  While a grouped launch is open (begin_group) the sum is collected and runs with the
        others in ONE lvsr_colsum_many launch at flush_group() — X must still hold its values then, `out` must not be read before
        (the bias gradients of a backward pass: 21 launches of 5-13 us per training step before round 6)."""
        M = X.shape[0] if M is None else M
        N = X.shape[1] if N is None else N
        ldx = X.stride(0) if ldx is None else ldx
        pending = getattr(self, "_group_colsums", None)
        if pending is not None and getattr(self, "_group", None) is not None:
            pending.append((X, out, float(beta), int(M), int(N), int(ldx), (ws.numel() * 4 if ws is not None else 0)))
            return
        self.call("lvsr_colsum", self.stream_for(out), ptr(X), M, N, ldx, ptr(out), beta,
                  ptr(ws), (ws.numel() * 4 if ws is not None else 0))

    def _flush_colsums(self, ws):
        pending, self._group_colsums = getattr(self, "_group_colsums", None), None
        if not pending:
            return
        outs = {p[1].data_ptr() for p in pending}
        split_ws = {p[6] for p in pending}
        if len(outs) != len(pending) or len(split_ws) != 1 or len(pending) > 32:
            for X, out, beta, M, N, ldx, wsb in pending:       # (a shared output or mixed workspaces: as separate launches, in order)
                self.call("lvsr_colsum", self.stream_for(out), ptr(X), M, N, ldx, ptr(out), beta, ptr(ws) if wsb else None, min(wsb, ws.numel() * 4))
            return
        cls = self.structs["lvsr_colsum_desc"]
        arr = (cls * len(pending))()
        for d, (X, out, beta, M, N, ldx, wsb) in zip(arr, pending):
            d.X, d.out, d.M, d.N, d.ldx, d.beta = X.data_ptr(), out.data_ptr(), M, N, ldx, beta
        wsb = split_ws.pop()
        self.call("lvsr_colsum_many", self.stream_for(pending[0][1]), arr, len(pending), ptr(ws) if wsb else None, ws.numel() * 4 if wsb else 0, wsb)

    def copy_many(self, pairs):
        """pairs: [(src, dst)] or [(src, dst, beta)] of equally shaped 1-D / 2-D fp32 tensors with unit inner stride -> one
        lvsr_copy2d_many launch (per 32 pairs); beta != 0: dst = src + beta * dst."""
        if not pairs:
            return
        cls = self.structs["lvsr_copy_desc"]
        arr = (cls * len(pairs))()
        for d, pair in zip(arr, pairs):
            src, dst = pair[0], pair[1]
            d.beta = float(pair[2]) if len(pair) > 2 else 0.0
            assert src.shape == dst.shape and src.dim() in (1, 2) and src.stride(-1) == 1 and dst.stride(-1) == 1
            rows, cols = (1, src.shape[0]) if src.dim() == 1 else (src.shape[0], src.shape[1])
            d.src, d.dst, d.rows, d.cols = src.data_ptr(), dst.data_ptr(), rows, cols
            d.lds = src.stride(0) if src.dim() == 2 else cols
            d.ldd = dst.stride(0) if dst.dim() == 2 else cols
        self.call("lvsr_copy2d_many", self.stream_for(pairs[0][1]), arr, len(pairs))

    def transpose(self, x, out):
        self.call("lvsr_transpose", self.stream_for(out), ptr(x), x.shape[0], x.shape[1], ptr(out))

    def pack_size(self, K, N):
        return int(self._lvsr_pack_size(K, N))

    def pack_b(self, W, packed, trans=False):
        """W (K,N) [or (N,K) with trans=True] -> packed operand copy for the step kernels."""
        K, N = (W.shape[1], W.shape[0]) if trans else (W.shape[0], W.shape[1])
        assert packed.numel() >= self.pack_size(K, N)
        self.call("lvsr_pack_b", self.stream_for(packed), ptr(W), W.stride(0), K, N, int(trans), ptr(packed))

    def pack_many(self, jobs, use_graph=False, cache=None):
        """jobs: [(W 2-D tensor, packed 1-D tensor, trans)] -> one lvsr_pack_b_many call.  `cache` (a dict owned by the caller)
        keeps the descriptor array between calls while the tensors stay where they are."""
        if not jobs:
            return
        sig = tuple((W.data_ptr(), W.stride(0), W.shape[0], W.shape[1], P.data_ptr(), int(t)) for W, P, t in jobs)
        arr = None if cache is None or cache.get("sig") != sig else cache.get("arr")
        if arr is None:
            cls = self.structs["lvsr_pack_desc"]
            arr = (cls * len(jobs))()
            for d, (W, P, t) in zip(arr, jobs):
                K, N = (W.shape[1], W.shape[0]) if t else (W.shape[0], W.shape[1])
                assert W.stride(1) == 1 and P.numel() >= self.pack_size(K, N)
                d.W, d.packed, d.ldw, d.K, d.N, d.trans = W.data_ptr(), P.data_ptr(), W.stride(0), K, N, int(t)
            if cache is not None:
                cache["sig"], cache["arr"] = sig, arr
        ref = jobs[0][1]
        self.call("lvsr_pack_b_many", self.stream_for(ref), arr, len(jobs), int(bool(use_graph) and ref.is_cuda))

    def run(self, fn, struct_name, ref_tensor, use_graph=None, **fields):
        """Call an args-struct entry point: fn(stream, &args[, use_graph])."""
        a = self.make(struct_name, **fields)
        if use_graph is None:
            self.call(fn, self.stream_for(ref_tensor), ctypes.byref(a))
        else:
            self.call(fn, self.stream_for(ref_tensor), ctypes.byref(a), int(use_graph))
        if use_graph:
            self.after_graph(ref_tensor, max(fields.get("T", 0), fields.get("L", 0)))
        return a

    def after_graph(self, ref_tensor, steps):
        """Block the host until a long time-loop graph has drained (see the measurement at the top of this file)."""
        if self.sync_after_graph and ref_tensor.is_cuda and steps >= 16 and not self.capturing:
            torch.cuda.current_stream(ref_tensor.device).synchronize()

    def region(self, owner, key, ref_tensor, enabled=True, volatile=(), drain=True):
        """`owner`: the object whose buffers the region's launches reference (it carries the bookkeeping, so it dies with
        it, and a process-unique token, so a later object at a recycled address can never hit its graphs)."""
        return Region(self, owner, key, ref_tensor, enabled, volatile, drain)

    _uid = [0]

    @classmethod
    def unique_token(cls):
        cls._uid[0] += 1
        return cls._uid[0]


class Region(object):
    """One cached hipGraph for everything the host enqueues between `begin()` and `end()`:

        out = lib.region(key, tensor).run(fn)

    where fn() enqueues (library calls, tensor copies/fills on the current stream) and returns whatever the caller needs
    back on a replay (tensors living in stable workspaces).

    `key` + `volatile` must name every pointer, shape and scalar the enqueue code depends on.  The first time a `key` is
    seen the body runs eagerly (so every workspace exists), the second time it is captured, afterwards replayed; a changed
    `volatile` part (workspace generation: buffers were re-allocated) re-captures without another eager pass.  A capture during
    which the caching allocator handed out memory is dropped (a replay would write to memory it does not own) and the key
    is enqueued again eagerly, and stays eager.  Disabled on the emulator / CPU tensors, with `lib.step_graph = False`, and inside another region."""
    def __init__(self, lib, owner, key, ref, enabled, volatile=(), drain=True):
        self.lib, self.ref = lib, ref
        self.drain = drain          # block the host until a replay has drained (long time-loop regions); False: short regions
        if not hasattr(owner, "_region_token"):
            owner._region_token, owner._regions = lib.unique_token(), {}
        soft = repr((owner._region_token, key)).encode()
        self.kb = soft + b"|" + repr(volatile).encode()
        self.enabled = bool(enabled) and lib.step_graph and ref.is_cuda and not lib.is_emulator and not lib.capturing
        self.state = "eager"
        self.slot = owner._regions.setdefault(soft, dict(seen=0, result=None)) if self.enabled else dict(seen=0, result=None)

    @property
    def result(self):
        return self.slot["result"]

    @result.setter
    def result(self, v):
        self.slot["result"] = v

    def _alloc_count(self):
        return torch.cuda.memory_stats(self.ref.device).get("allocation.all.allocated", 0)

    def begin(self):
        if not self.enabled:
            return True
        self.slot["seen"] += 1
        if self.slot.get("bad"):
            return True
        if self.slot["seen"] == 1:
            # the eager pass that lets every workspace come into being: its inner time-loop graphs would never be replayed
            # (the region is captured next time), so do not build them
            self.state = "first"
            self.lib._lvsr_graph_suppress(1)
            return True
        lib = self.lib
        rc = lib._lvsr_region_begin(lib.stream_for(self.ref), self.kb, len(self.kb))
        if rc < 0:
            raise NativeError("lvsr_region_begin failed (%d): %s" % (rc, lib.last_error()))
        if rc == 1:
            self.state = "replayed"
            if self.drain:
                lib.after_graph(self.ref, 1 << 20)
            return False
        if rc == 2:
            self.slot["bad"] = True
            return True
        self.state = "capturing"
        lib.capturing = True
        self._allocs = self._alloc_count()
        return True

    def end(self):
        if self.state == "first":
            self.lib._lvsr_graph_suppress(0)
            self.state = "eager"
            return True
        if self.state != "capturing":
            return True
        lib = self.lib
        lib.capturing = False
        clean = self._alloc_count() == self._allocs
        rc = lib._lvsr_region_end(lib.stream_for(self.ref), int(clean))
        self.state = "captured"
        if rc != 0:
            self.slot["bad"] = True
            raise NativeError("lvsr_region_end failed (%d): %s" % (rc, lib.last_error()))
        if not clean:
            self.slot["bad"] = True
            return False
        if self.drain:
            lib.after_graph(self.ref, 1 << 20)
        return True

    def run(self, fn):
        if self.begin():
            ok = True
            try:
                self.result = fn()
            finally:
                ok = self.end()
            if not ok:
                self.result = fn()
        return self.result


_default = None


def get():
    """The product library (gfx950).  Raises if it is not built."""
    global _default
    if _default is None:
        _default = Lib(DEFAULT_LIB)
    return _default
