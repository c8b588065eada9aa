"""FST language model + shallow fusion for beam-search decoding (A17 of SURVEY.md §8a).

What the reference's Theano host Ops compute (lvsr/ops.py:37-225: `FST.transition`, `FST.expand` = epsilon closure in the log
semiring, `FSTTransitionOp`, `FSTCostsOp`; MAX_STATES = 7 padded state sets) is computed here by `CsrWalk` over the same CSR arc
tables the device kernel walks, wrapped like `FSTTransition` / `LanguageModel` (lvsr/bricks/language_models.py:14-72, 107-137).  The fusion itself (`ShallowFusionReadout.readout`, language_models.py:92-104) is the device kernel
`lvsr_shallow_fusion`.  The reference reads OpenFST binaries through PyFST; here the automaton is an in-memory arc list
with an AT&T text reader and a reader/writer for OpenFST's binary `vector` / `standard` container (`read_openfst_binary`).

`DeviceFSTLanguageModel` keeps the per-hypothesis state sets in device memory and does the walk in the
HIP kernel `lvsr_fst_lm_step` over a CSR arc table, with the same interface, so beam search uses either unchanged.
"""
import math
import struct
from collections import defaultdict

import numpy
import torch

EPSILON = 0          # lvsr/ops.py:22-24
MAX_STATES = 7
NOT_STATE = -1


def _toposort_flatten(depends):
    """toposort.toposort_flatten(dict_of_sets): Kahn levels, each level sorted."""
    data = {k: set(v) for k, v in depends.items()}
    for k, v in data.items():
        v.discard(k)
    extra = set()
    for v in data.values():
        extra |= v
    for e in extra - set(data):
        data[e] = set()
    out = []
    while True:
        ready = sorted(k for k, v in data.items() if not v)
        if not ready:
            break
        out.extend(ready)
        data = {k: v - set(ready) for k, v in data.items() if k not in ready}
    if data:
        raise ValueError("cyclic epsilon dependencies in the FST")
    return out


class ArcFST(object):
    """Weighted acceptor in the tropical/log convention of OpenFST text files: weights are -log probabilities."""
    def __init__(self, start=0):
        self.start = start
        self.arcs = defaultdict(list)          # state -> [(ilabel, nextstate, weight)]
        self.final = {}
        self.isyms = {}

    def add_arc(self, src, dst, ilabel, weight=0.0):
        self.arcs[src].append((int(ilabel), int(dst), float(weight)))

    @classmethod
    def from_att_text(cls, lines, isyms=None):
        """AT&T text format (`fstprint`): 'src dst ilabel [olabel] [weight]' arcs, 'state [weight]' finals; the first
        source state is the start state.  `isyms` maps symbol strings to integer labels (`<eps>` = 0)."""
        f, first = cls(), True
        for line in lines:
            p = line.split()
            if not p:
                continue
            if len(p) <= 2:
                f.final[int(p[0])] = float(p[1]) if len(p) == 2 else 0.0
                continue
            src, dst = int(p[0]), int(p[1])
            lab = p[2]
            il = isyms[lab] if isyms is not None and not lab.lstrip("-").isdigit() else int(lab)
            w = 0.0
            if len(p) == 4:
                try:
                    w = float(p[3]) if not p[3].lstrip("-").isdigit() or isyms is None else 0.0
                except ValueError:
                    w = 0.0
            elif len(p) >= 5:
                w = float(p[4])
            if first:
                f.start, first = src, False
            f.add_arc(src, dst, il, w)
        if isyms is not None:
            f.isyms = dict(isyms)
        return f

    # ---- the state-set walk (what lvsr/ops.py:51-97 computes), over the CSR tables the device kernel uses ---------------
    def walker(self):
        """The host walk over this automaton's CSR arc tables (built once; invalidated when an arc is added)."""
        sig = sum(len(v) for v in self.arcs.values())
        if getattr(self, "_walker", None) is None or self._walker_sig != sig:
            self._walker, self._walker_sig = CsrWalk(build_fst_table(self, {}, 0)), sig
        return self._walker

    def transition(self, states, character):
        """{state: -log weight} -> the set reached by arcs labelled `character` (no epsilon closure); parallel arcs and
        several sources of one target are combined in the log semiring."""
        return self.walker().advance(states, character)

    def expand(self, states):
        """Epsilon closure of a weighted state set in the log semiring."""
        return self.walker().close(states)


def log_add(terms):
    """-log(sum_i exp(-x_i)) of a non-empty list of -log weights, shifted by the largest term like the reference's
    combine_weights (lvsr/ops.py:51-54) so that equal inputs give equal float64 results."""
    m = max(terms)
    return m - math.log(sum(math.exp(m - x) for x in terms))


class CsrWalk(object):
    """Host-side walk of a weighted state set over the CSR arc tables of `build_fst_table` — the same tables, topological
    ranks and order of operations as the device kernel `lvsr_fst_lm_step` (csrc/fst_lm.hip), so the two are each other's
    check; both are pinned to the reference's own dict walk through tests/golden/fst_walk.npz.
      advance: labelled arcs of a state are sorted by label, so the arcs of one character are a contiguous run found by
               bisection; contributions to one target are collected in (source order, arc order) and log-added;
      close:   states reachable over epsilon arcs are relaxed in increasing topological rank; a state's weight is final when
               it is visited (all its epsilon predecessors rank lower), then pushed along its epsilon arcs."""
    def __init__(self, table):
        self.t = table
        self.arc_off, self.arc_lab = table["arc_off"].tolist(), table["arc_lab"].tolist()
        self.arc_dst, self.arc_w = table["arc_dst"].tolist(), table["arc_w"].tolist()
        self.eps_off, self.eps_dst, self.eps_w = table["eps_off"].tolist(), table["eps_dst"].tolist(), table["eps_w"].tolist()
        self.rank = table["topo"].tolist()
        self.num_states = table["num_states"]

    def advance(self, states, label):
        import bisect
        into = {}
        for q, w in states.items():
            if q < 0 or q >= self.num_states:
                continue
            lo, hi = self.arc_off[q], self.arc_off[q + 1]
            i = bisect.bisect_left(self.arc_lab, label, lo, hi)
            while i < hi and self.arc_lab[i] == label:
                into.setdefault(self.arc_dst[i], []).append(w + self.arc_w[i])
                i += 1
        return {d: log_add(terms) for d, terms in into.items()}

    def close(self, states):
        out = dict(states)
        reach, stack = set(out), list(out)
        while stack:                                   # everything reachable over epsilon arcs
            q = stack.pop()
            if q < 0 or q >= self.num_states:
                continue
            for i in range(self.eps_off[q], self.eps_off[q + 1]):
                d = self.eps_dst[i]
                if d not in reach:
                    reach.add(d)
                    stack.append(d)
        pending = {}
        for q in sorted((q for q in reach if 0 <= q < self.num_states), key=lambda q_: self.rank[q_]):
            terms = pending.pop(q, None)
            if terms:
                out[q] = log_add(([out[q]] if q in out else []) + terms)
            w = out[q]
            for i in range(self.eps_off[q], self.eps_off[q + 1]):
                pending.setdefault(self.eps_dst[i], []).append(w + self.eps_w[i])
        return out


def _pad(arr, value):
    arr = list(arr)
    if len(arr) > MAX_STATES:
        raise ValueError("FST state set larger than MAX_STATES=%d (lvsr/ops.py:23,140-142)" % MAX_STATES)
    return numpy.asarray(arr + [value] * (MAX_STATES - len(arr)))


class FSTLanguageModel(object):
    """`LanguageModel` + `FSTTransition` (language_models.py:14-72,107-137) with the fusion settings of
    `SpeechRecognizer.__init__` (recognizer.py:322-337: weight, normalize_am_weights=True, normalize_lm_weights=False,
    normalize_tot_weights=False, am_beta=1.0 defaults)."""
    def __init__(self, fst, nn_char_map=None, remap_table=None, no_transition_cost=1e12, weight=0.0,
                 normalize_am_weights=True, normalize_lm_weights=False, normalize_tot_weights=False, am_beta=1.0):
        self.fst = fst
        if remap_table is None:
            fst_char_map = {k: v for k, v in fst.isyms.items() if k != "<eps>"}
            if len(fst_char_map) != len(nn_char_map):
                raise ValueError()                                            # language_models.py:116-117
            remap_table = {nn_char_map[ch]: code for ch, code in fst_char_map.items()}
        self.remap_table = dict(remap_table)
        self.no_transition_cost = no_transition_cost
        self.lm_weight, self.am_beta = float(weight), float(am_beta)
        self.norm = (bool(normalize_am_weights), bool(normalize_lm_weights), bool(normalize_tot_weights))
        self.out_dim = len(self.remap_table)
        self.device_add = None
        # memo of the walk on EXACT keys (state ids and float64 weights as they are): a hit returns precisely what the walk
        # would compute again, so decoding stays a pure function of its inputs (bit-for-bit reproducible beams)
        self._cost_cache = {}
        self._trans_cache = {}

    @staticmethod
    def _key(sd):
        return tuple(sorted(sd.items()))

    # FSTCostsOp.perform, lvsr/ops.py:206-225
    def costs(self, states, weights):
        out = []
        for st, wt in zip(states, weights):
            sd = dict(zip(st.tolist(), wt.tolist()))
            sd.pop(NOT_STATE, None)
            key = self._key(sd)
            c = self._cost_cache.get(key)
            if c is None:
                c = numpy.ones(self.out_dim, dtype=numpy.float32) * self.no_transition_cost
                if sd:
                    total = log_add(list(sd.values()))
                    for nn_ch, fst_ch in self.remap_table.items():
                        nxt = self.fst.expand(self.fst.transition(sd, fst_ch))
                        if nxt:
                            c[nn_ch] = log_add(list(nxt.values())) - total
                if len(self._cost_cache) > 100000:
                    self._cost_cache.clear()
                self._cost_cache[key] = c
            out.append(c)
        return numpy.array(out, dtype=numpy.float32).reshape(len(states), self.out_dim)

    def initial_states(self, n):
        """FSTTransition.initial_states, language_models.py:52-62."""
        sd = self.fst.expand({self.fst.start: 0.0})
        states = numpy.tile(_pad(sd.keys(), NOT_STATE).astype(numpy.int64)[None, :], (n, 1))
        weights = numpy.tile(_pad(sd.values(), 0).astype(numpy.float64)[None, :], (n, 1))
        return dict(states=states, weights=weights, add=self.costs(states, weights))

    def transition(self, lm_states, outputs):
        """FSTTransitionOp.perform (lvsr/ops.py:147-169) then FSTCostsOp on the new states (language_models.py:40-50)."""
        ns, nw = [], []
        for st, wt, ch in zip(lm_states["states"], lm_states["weights"], outputs):
            sd = dict(zip(st.tolist(), wt.tolist()))
            sd.pop(NOT_STATE, None)
            ck = (self._key(sd), int(ch))
            hit = self._trans_cache.get(ck)
            if hit is None:
                nxt = self.fst.expand(self.fst.transition(sd, self.remap_table[int(ch)]))
                hit = (list(nxt.keys()), list(nxt.values()))
                if len(self._trans_cache) > 100000:
                    self._trans_cache.clear()
                self._trans_cache[ck] = hit
            ns.append(_pad(hit[0], NOT_STATE))
            nw.append(_pad(hit[1], 0))
        states = numpy.array(ns, dtype=numpy.int64).reshape(len(outputs), MAX_STATES)
        weights = numpy.array(nw, dtype=numpy.float64).reshape(len(outputs), MAX_STATES)
        return dict(states=states, weights=weights, add=self.costs(states, weights))

    @staticmethod
    def take(lm_states, indexes):
        return {k: numpy.take(v, indexes, axis=0) for k, v in lm_states.items()}

    def stage(self, lm_states, device=None):
        """Upload `lm_add` (n,V) for the fusion kernel."""
        self.device_add = torch.from_numpy(numpy.ascontiguousarray(lm_states["add"], dtype=numpy.float32))
        if device is not None:
            self.device_add = self.device_add.to(device)
        return self.device_add


class DeviceFSTLanguageModel(FSTLanguageModel):
    """FSTLanguageModel whose state sets live on the device: `transition` + look-ahead costs are one launch of
    `lvsr_fst_lm_step` per beam step (no Python walk, no host<->device traffic except the 4-byte error word)."""
    def __init__(self, fst, device, lib=None, **kw):
        super(DeviceFSTLanguageModel, self).__init__(fst, **kw)
        from . import native
        self.lib = lib if lib is not None else native.get()
        self.device = torch.device(device)
        self.table = build_fst_table(fst, self.remap_table, self.out_dim)
        self._dev = {k: torch.from_numpy(v).to(self.device) for k, v in self.table.items() if isinstance(v, numpy.ndarray)}
        d = self._dev
        self._fst = self.lib.make("lvsr_fst", arc_off=d["arc_off"], arc_lab=d["arc_lab"], arc_dst=d["arc_dst"],
                                  arc_w=d["arc_w"], eps_off=d["eps_off"], eps_dst=d["eps_dst"], eps_w=d["eps_w"],
                                  topo=d["topo"], remap=d["remap"], num_states=self.table["num_states"], V=self.out_dim,
                                  no_transition_cost=float(self.no_transition_cost))
        self._err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._initial = {}
        start = self.fst.expand({self.fst.start: 0.0})
        self._start = (_pad(start.keys(), NOT_STATE).astype(numpy.int64), _pad(start.values(), 0).astype(numpy.float64))

    _ERRORS = {1: "FST state set larger than MAX_STATES=%d (lvsr/ops.py:23,140-142)" % MAX_STATES,
               2: "FST candidate set outgrew the device kernel's capacity (16 states)",
               3: "chosen character has no FST input label"}

    def _step(self, states, weights, outputs):
        import ctypes
        from .native import ptr
        n = states.shape[0]
        add = torch.empty((n, self.out_dim), dtype=torch.float32, device=self.device)
        if outputs is None:
            ns, nw = states, weights
        else:
            ns, nw = torch.empty_like(states), torch.empty_like(weights)
        if n:
            self.lib.call("lvsr_fst_lm_step", self.lib.stream_for(states), ctypes.byref(self._fst), ptr(states), ptr(weights),
                          ptr(outputs), n, ptr(ns) if outputs is not None else None,
                          ptr(nw) if outputs is not None else None, ptr(add), ptr(self._err))
            code = int(self._err.item())
            if code:
                self._err.zero_()
                raise ValueError(self._ERRORS.get(code, "lvsr_fst_lm_step error %d" % code))
        return dict(states=ns, weights=nw, add=add)

    def check_error(self):
        """Raise what the device walk flagged since the last check (the beam-search driver looks once per search: the error
        word is sticky and the kernels of a flagged row keep producing finite numbers)."""
        code = int(self._err.item())
        if code:
            self._err.zero_()
            raise ValueError(self._ERRORS.get(code, "lvsr_fst_lm_step error %d" % code))

    def initial_states(self, n):
        """State sets / look-ahead costs of n fresh hypotheses.  Computed once per n and kept (callers copy them into their own
        buffers): starting a search then costs no launch and, unlike `_step`, no look at the error word."""
        hit = self._initial.get(n)
        if hit is None:
            st = torch.from_numpy(numpy.tile(self._start[0][None, :], (n, 1))).to(self.device)
            wt = torch.from_numpy(numpy.tile(self._start[1][None, :], (n, 1))).to(self.device)
            hit = self._initial[n] = self._step(st, wt, None)
        return {k: v.clone() for k, v in hit.items()}

    def transition(self, lm_states, outputs):
        if torch.is_tensor(outputs):
            out = outputs.to(device=self.device, dtype=torch.int64).contiguous()
        else:
            out = torch.as_tensor(numpy.ascontiguousarray(outputs), dtype=torch.int64).to(self.device).contiguous()
        return self._step(lm_states["states"].contiguous(), lm_states["weights"].contiguous(), out)

    on_device = True

    def take(self, lm_states, indexes):
        idx = indexes if torch.is_tensor(indexes) else torch.as_tensor(numpy.ascontiguousarray(indexes), dtype=torch.int64).to(self.device)
        return {k: v.index_select(0, idx) for k, v in lm_states.items()}

    def stage(self, lm_states, device=None):
        self.device_add = lm_states["add"]
        return self.device_add


def build_fst_table(fst, remap_table, V):
    """CSR arc table for `lvsr_fst_lm_step` (include/lvsr_hip.h `lvsr_fst`)."""
    ids = {fst.start}
    for src, lst in fst.arcs.items():
        ids.add(src)
        ids.update(d for _, d, _ in lst)
    S = max(ids) + 1
    lab = [[] for _ in range(S)]
    eps = [[] for _ in range(S)]
    for src, lst in fst.arcs.items():
        for (il, d, w) in lst:
            (eps if il == EPSILON else lab)[src].append((il, d, w))
    for q in range(S):
        lab[q].sort(key=lambda a: a[0])                       # stable: arcs with equal labels keep their order
    order = _toposort_flatten(_eps_depends(eps))
    topo = numpy.zeros(S, dtype=numpy.int32)
    rank = {q: i for i, q in enumerate(order)}
    nxt = len(order)
    for q in range(S):
        if q in rank:
            topo[q] = rank[q]
        else:
            topo[q] = nxt
            nxt += 1

    def csr(rows):
        off = numpy.zeros(S + 1, dtype=numpy.int32)
        for q in range(S):
            off[q + 1] = off[q] + len(rows[q])
        flat = [a for r in rows for a in r]
        return (off, numpy.array([a[0] for a in flat] or [0], dtype=numpy.int32),
                numpy.array([a[1] for a in flat] or [0], dtype=numpy.int32),
                numpy.array([a[2] for a in flat] or [0.0], dtype=numpy.float64))
    arc_off, arc_lab, arc_dst, arc_w = csr(lab)
    eps_off, _, eps_dst, eps_w = csr(eps)
    remap = numpy.full(V, -1, dtype=numpy.int32)
    for nn_ch, fst_ch in remap_table.items():
        if 0 <= int(nn_ch) < V:
            remap[int(nn_ch)] = int(fst_ch)
    return dict(arc_off=arc_off, arc_lab=arc_lab, arc_dst=arc_dst, arc_w=arc_w, eps_off=eps_off, eps_dst=eps_dst,
                eps_w=eps_w, topo=topo, remap=remap, num_states=S)


def _eps_depends(eps):
    dep = defaultdict(set)
    for q, lst in enumerate(eps):
        for (_, d, _) in lst:
            dep[d].add(q)
    return dep                      # _toposort_flatten raises ValueError on an epsilon cycle


# ---- OpenFST binary container (`fstcompile` output; what PyFST's `fst.read` loads in lvsr/ops.py:40-41) -----------
# Layout (OpenFST src/include/fst/fst.h FstHeader::Write, vector-fst.h): int32 magic 2125659606; string fst type
# ("vector"); string arc type ("standard"); int32 version (2); int32 flags (bit0 has isymbols, bit1 has osymbols,
# bit2 aligned); uint64 properties; int64 start; int64 num_states; int64 num_arcs; [symbol tables]; then per state:
# float32 final weight, int64 narcs, narcs x {int32 ilabel, int32 olabel, float32 weight, int32 nextstate}.
# Strings are int32 length + bytes.  Symbol table: int32 magic 2125658996, string name, int64 available_key, int64 size,
# size x {string symbol, int64 key}.  Stated from the OpenFST sources' documented format; no OpenFST binary exists in the
# reference tree or this image: the reader is tested on its own writer's output and on a file assembled byte by byte in
# tests/test_lm.py from that layout (both symbol tables, an epsilon arc, 64-bit properties, kNoStateId headers) — not on a file
# OpenFST itself wrote.
_FST_MAGIC = 2125659606
_SYMTAB_MAGIC = 2125658996


def _rd(fmt, buf, pos):
    size = struct.calcsize(fmt)
    return struct.unpack_from(fmt, buf, pos), pos + size


def _rd_str(buf, pos):
    (n,), pos = _rd("<i", buf, pos)
    return buf[pos:pos + n].decode("utf-8"), pos + n


def _rd_symtab(buf, pos):
    (magic,), pos = _rd("<i", buf, pos)
    if magic != _SYMTAB_MAGIC:
        raise ValueError("bad OpenFST symbol table magic %d" % magic)
    _, pos = _rd_str(buf, pos)
    (_, size), pos = _rd("<qq", buf, pos)
    table = {}
    for _ in range(size):
        sym, pos = _rd_str(buf, pos)
        (key,), pos = _rd("<q", buf, pos)
        table[sym] = key
    return table, pos


def read_openfst_binary(path_or_bytes):
    """-> ArcFST from an OpenFST binary `vector` FST with `standard` (tropical, float32) arcs (what `fst.read` of lvsr/ops.py:44
    is given: the outputs of bin/lm2fst.sh).  Output labels are read over (the reference looks at `arc.ilabel` only, ops.py:57-59)."""
    buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    try:
        return _parse_openfst(buf)
    except struct.error:
        raise ValueError("truncated OpenFST binary (%d bytes)" % len(buf))


def _parse_openfst(buf):
    (magic,), pos = _rd("<i", buf, 0)
    if magic != _FST_MAGIC:
        raise ValueError("not an OpenFST binary (magic %d)" % magic)
    fst_type, pos = _rd_str(buf, pos)
    arc_type, pos = _rd_str(buf, pos)
    if fst_type != "vector" or arc_type != "standard":
        raise ValueError("unsupported OpenFST container %s/%s (need vector/standard; convert with fstconvert)"
                         % (fst_type, arc_type))
    (version, flags), pos = _rd("<ii", buf, pos)
    (_props, start, num_states, _num_arcs), pos = _rd("<Qqqq", buf, pos)
    if flags & 4:
        raise ValueError("aligned OpenFST files are not supported (write without --fst_align)")
    f = ArcFST(start=start)
    if flags & 1:
        f.isyms, pos = _rd_symtab(buf, pos)
    if flags & 2:
        _, pos = _rd_symtab(buf, pos)
    q = 0
    # num_states = -1 (kNoStateId): the header went out before the states were counted — they run to the end of the file
    while (q < num_states) if num_states >= 0 else (pos < len(buf)):
        (final, narcs), pos = _rd("<fq", buf, pos)
        if final != float("inf"):
            f.final[q] = final
        for _ in range(narcs):
            (il, _ol, w, nxt), pos = _rd("<iifi", buf, pos)
            f.add_arc(q, nxt, il, w)
        q += 1
    return f


def write_openfst_binary(fst, path=None):
    """Inverse of `read_openfst_binary` (acceptor: olabel = ilabel); returns the bytes and writes them if `path`."""
    def wstr(s_):
        b = s_.encode("utf-8")
        return struct.pack("<i", len(b)) + b
    ids = {fst.start} | set(fst.arcs) | {d for lst in fst.arcs.values() for (_, d, _) in lst} | set(fst.final)
    S = max(ids) + 1
    narcs = sum(len(v) for v in fst.arcs.values())
    out = [struct.pack("<i", _FST_MAGIC), wstr("vector"), wstr("standard"), struct.pack("<ii", 2, 1 if fst.isyms else 0),
           struct.pack("<Qqqq", 0, fst.start, S, narcs)]
    if fst.isyms:
        out += [struct.pack("<i", _SYMTAB_MAGIC), wstr("isyms"), struct.pack("<qq", max(fst.isyms.values()) + 1, len(fst.isyms))]
        for sym, key in sorted(fst.isyms.items(), key=lambda kv: kv[1]):
            out += [wstr(sym), struct.pack("<q", key)]
    for q in range(S):
        arcs = fst.arcs.get(q, [])
        out.append(struct.pack("<fq", fst.final.get(q, float("inf")), len(arcs)))
        for (il, d, w) in arcs:
            out.append(struct.pack("<iifi", il, il, w, d))
    data = b"".join(out)
    if path:
        with open(path, "wb") as fh:
            fh.write(data)
    return data


def language_model_from_config(lm, character_map, device, lib=None):
    """The `lm:` block of the reference's `net:` section (lvsr/bricks/recognizer.py:322-337; LanguageModel.__init__,
    language_models.py:107-121): {path, weight, no_transition_cost, normalize_am_weights, normalize_lm_weights,
    normalize_tot_weights, am_beta}.  `path` is an OpenFST binary (vector/standard, with input symbols) or an AT&T text
    file next to `<path>.isyms` ('symbol id' lines).  `device_walk: false` (ours) selects the host walk."""
    lm = dict(lm)
    path = lm.pop("path")
    device_walk = lm.pop("device_walk", True)
    lm.pop("type_", None)
    with open(path, "rb") as fh:
        head = fh.read(4)
    if head == struct.pack("<i", _FST_MAGIC):
        fst = read_openfst_binary(path)
    else:
        with open(path + ".isyms") as fh:
            isyms = {p[0]: int(p[1]) for p in (line.split() for line in fh) if len(p) == 2}
        with open(path) as fh:
            fst = ArcFST.from_att_text(fh.read().splitlines(), isyms)
    if not fst.isyms:
        raise ValueError("the FST at %s carries no input symbol table" % path)
    if device_walk:
        return DeviceFSTLanguageModel(fst, device, lib=lib, nn_char_map=character_map, **lm)
    return FSTLanguageModel(fst, nn_char_map=character_map, **lm)


def char_ngram_fst(num_chars, seed=0, order=2, eps_backoff=True):
    """Deterministic synthetic character LM (BASELINE.json config 'WSJ decode' asks for a synthetic FST LM): states =
    previous character (+ a unigram back-off state reached by an epsilon arc), random but fixed -log probabilities."""
    rng = numpy.random.RandomState(seed)
    f = ArcFST(start=0)
    f.isyms = {"<eps>": 0}
    f.isyms.update({"c%d" % i: i + 1 for i in range(num_chars)})
    backoff = num_chars + 1
    uni = rng.dirichlet(numpy.ones(num_chars) * 2.0)
    for c in range(num_chars):
        f.add_arc(backoff, 1 + c, c + 1, -math.log(uni[c]))
    for s in [0] + list(range(1, num_chars + 1)):
        keep = rng.choice(num_chars, size=max(2, num_chars // 2), replace=False)
        p = rng.dirichlet(numpy.ones(len(keep)))
        for c, pc in zip(keep, p):
            f.add_arc(s, 1 + int(c), int(c) + 1, -math.log(0.8 * pc))
        if eps_backoff:
            f.add_arc(s, backoff, EPSILON, -math.log(0.2))
    return f, {"c%d" % i: i for i in range(num_chars)}
