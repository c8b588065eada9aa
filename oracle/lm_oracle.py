"""CPU ORACLE of the FST language model walk and shallow fusion.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Parity status of the FST walk: PINNED to the reference's own code — `oracle/theano_harness/gen_fst_golden.py` runs
lvsr/ops.py (FST.transition / FST.expand / FSTTransitionOp.perform / FSTCostsOp.perform) from the scratch copy and records
state sets, weights and look-ahead costs along random walks in tests/golden/fst_walk.npz; this oracle, the product's host
walk and its device kernel are all checked against that fixture (tests/test_lm.py).  Only the automaton CONTAINER is a
stand-in there (PyFST / OpenFST cannot be installed: an AT&T-text parser exposing the attributes ops.py touches), so
reading OpenFST binary files remains unpinned.  This oracle is an INDEPENDENT dense formulation of the same semantics (log-semiring matrices instead of dict walks): state sets are weight vectors over all states,
`transition` = vector (x) arc-matrix of one label, `expand` = epsilon closure by repeated relaxation; costs as in
FSTCostsOp (:206-225).  Fusion follows ShallowFusionReadout.readout (lvsr/bricks/language_models.py:92-104) + LMEmitter
and is PINNED too: the same fixture holds the reference bricks' output (evaluated by Theano) for all 8 normalisation flag
combinations x 2 weightings.
"""
import numpy

INF = numpy.inf


def logadd(a, b):
    m = numpy.minimum(a, b)
    with numpy.errstate(invalid="ignore", over="ignore"):
        r = m - numpy.log(numpy.exp(m - a) + numpy.exp(m - b))
    return numpy.where(numpy.isinf(m), m, r)


class DenseFST(object):
    def __init__(self, arcs, start, num_labels):
        """arcs: iterable of (src, dst, ilabel, weight); weights are -log probabilities."""
        n = 1 + max(max(a[0], a[1]) for a in arcs)
        self.n, self.start = n, start
        self.M = numpy.full((num_labels + 1, n, n), INF)
        for s, d, l, w in arcs:
            self.M[l, s, d] = logadd(self.M[l, s, d], w)

    def apply(self, vec, label):
        out = numpy.full(self.n, INF)
        for s in range(self.n):
            if numpy.isfinite(vec[s]):
                out = logadd(out, vec[s] + self.M[label, s])
        return out

    def expand(self, vec):
        """epsilon closure: the reference relaxes each epsilon-reachable state once in topological order, which for an
        acyclic epsilon graph equals the sum over all epsilon paths."""
        total = vec.copy()
        frontier = vec.copy()
        for _ in range(self.n):
            frontier = self.apply(frontier, 0)
            if not numpy.isfinite(frontier).any():
                break
            total = logadd(total, frontier)
        return total

    def initial(self):
        v = numpy.full(self.n, INF)
        v[self.start] = 0.0
        return self.expand(v)

    def step(self, vec, label):
        return self.expand(self.apply(vec, label))

    @staticmethod
    def total(vec):
        f = vec[numpy.isfinite(vec)]
        if len(f) == 0:
            return None
        m = f.min()
        return m - numpy.log(numpy.exp(m - f).sum())

    def costs(self, vec, remap, no_transition_cost):
        c = numpy.ones(len(remap), numpy.float32) * no_transition_cost
        tot = self.total(vec)
        if tot is not None:
            for nn, lab in remap.items():
                nt = self.total(self.step(vec, lab))
                if nt is not None:
                    c[nn] = nt - tot
        return c


def log_softmax(x):
    x = numpy.asarray(x, numpy.float64)
    m = x.max(axis=-1, keepdims=True)
    return x - m - numpy.log(numpy.exp(x - m).sum(axis=-1, keepdims=True))


def shallow_fusion(am, lm_add, lm_weight, am_beta=1.0, normalize_am=True, normalize_lm=False, normalize_tot=False):
    lm = -numpy.asarray(lm_add, numpy.float64)
    if normalize_lm:
        lm = log_softmax(lm)
    a = am_beta * numpy.asarray(am, numpy.float64)
    if normalize_am:
        a = log_softmax(a)
    x = a + lm_weight * lm
    if normalize_tot:
        x = log_softmax(x)
    return x
