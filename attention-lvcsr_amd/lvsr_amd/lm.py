"""FST language model + shallow fusion for beam-search decoding (A17 of SURVEY.md §8a).

Host-side state-set walk exactly as the reference's Theano host Ops do it (lvsr/ops.py:37-225: `FST.transition`,
`FST.expand` = epsilon closure in the log semiring with a toposorted relaxation, `FSTTransitionOp`, `FSTCostsOp`;
MAX_STATES = 7 padded state sets) wrapped like `FSTTransition` / `LanguageModel` (lvsr/bricks/language_models.py:14-72,
107-137).  The fusion itself (`ShallowFusionReadout.readout`, language_models.py:92-104) is the device kernel
`lvsr_shallow_fusion`.  The reference reads OpenFST binaries through PyFST; here the automaton is an in-memory arc list
(AT&T text format reader included) — an OpenFST binary reader is SURVEY.md §8f N4.
"""
import math
from collections import defaultdict, deque

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

    # ---- lvsr/ops.py:51-97 --------------------------------------------------------------------------
    @staticmethod
    def combine_weights(*args):
        m = max(a for a in args if a is not None) if any(a is not None for a in args) else None
        return m - math.log(sum(math.exp(m - x) for x in args if x is not None))

    def get_arcs(self, state, character):
        return [(state, nxt, il, w) for (il, nxt, w) in self.arcs.get(state, ()) if il == character]

    def transition(self, states, character):
        arcs = [a for state in states for a in self.get_arcs(state, character)]
        next_states = {}
        for next_state in {arc[1] for arc in arcs}:
            next_states[next_state] = self.combine_weights(*[states[arc[0]] + arc[3] for arc in arcs if arc[1] == next_state])
        return next_states

    def expand(self, states):
        seen, depends, queue = set(), defaultdict(list), deque()
        for state in states:
            queue.append(state)
            seen.add(state)
        while len(queue):
            state = queue.popleft()
            for arc in self.get_arcs(state, EPSILON):
                depends[arc[1]].append((arc[0], arc[3]))
                if arc[1] in seen:
                    continue
                queue.append(arc[1])
                seen.add(arc[1])
        order = _toposort_flatten({key: {s for s, _ in value} for key, value in depends.items()})
        next_states = states
        for next_state in order:
            next_states[next_state] = self.combine_weights(
                *([next_states.get(next_state)] + [next_states[prev] + weight for prev, weight in depends[next_state]]))
        return next_states


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
        # The walk is a pure function of the state set and of the weights RELATIVE to their minimum (costs are differences
        # of log-sums, so a common offset cancels; transitions carry the offset through).  Hypotheses of a beam share LM
        # states all the time, so memoising on that key removes almost all of the Python dict walking.
        self._cost_cache = {}
        self._trans_cache = {}

    @staticmethod
    def _key(sd):
        if not sd:
            return (), 0.0
        items = sorted(sd.items())
        base = min(w for _, w in items)
        return tuple((s, round(w - base, 10)) for s, w in items), base

    # FSTCostsOp.perform, lvsr/ops.py:206-225
    def costs(self, states, weights):
        out = []
        for st, wt in zip(states, weights):
            sd = dict(zip(st.tolist(), wt.tolist()))
            sd.pop(NOT_STATE, None)
            key, _ = self._key(sd)
            c = self._cost_cache.get(key)
            if c is None:
                c = numpy.ones(self.out_dim, dtype=numpy.float32) * self.no_transition_cost
                if sd:
                    total = self.fst.combine_weights(*sd.values())
                    for nn_ch, fst_ch in self.remap_table.items():
                        nxt = self.fst.expand(self.fst.transition(sd, fst_ch))
                        if nxt:
                            c[nn_ch] = self.fst.combine_weights(*nxt.values()) - total
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
            key, base = self._key(sd)
            ck = (key, int(ch))
            hit = self._trans_cache.get(ck)
            if hit is None:
                rel = {s_: w_ - base for s_, w_ in sd.items()}
                nxt = self.fst.expand(self.fst.transition(rel, self.remap_table[int(ch)]))
                hit = (list(nxt.keys()), list(nxt.values()))
                self._trans_cache[ck] = hit
            ns.append(_pad(hit[0], NOT_STATE))
            nw.append(_pad([w_ + base for w_ in hit[1]], 0))
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
