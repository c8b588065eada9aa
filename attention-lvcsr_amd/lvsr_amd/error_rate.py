"""Edit distance / error rates for decode reports (reference: lvsr/error_rate.py:11-76).
Host-side integer DP: not a kernel candidate (one hypothesis pair per utterance)."""
import numpy

COPY, INSERTION, DELETION, SUBSTITUTION = 0, 1, 2, 3      # lvsr/error_rate.py:3-6
INFINITY = 10 ** 9


def _edit_distance_matrix(y, y_hat):
    """dist[i, j] = edit distance between y[:i] and y_hat[:j]; action[i, j] = last action of an optimal chain
    (lvsr/error_rate.py:11-55; ties resolved in the reference's order: insertion, deletion, substitution, copy — the later
    match wins, and an insertion inherits the action of the cell above)."""
    n, m = len(y), len(y_hat)
    dist = numpy.zeros((n + 1, m + 1), dtype="int64")
    action = dist.copy()
    dist[:, 0] = numpy.arange(n + 1)
    dist[0, :] = numpy.arange(m + 1)
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            differ = y[i - 1] != y_hat[j - 1]
            insertion = dist[i - 1, j] + 1
            deletion = dist[i, j - 1] + 1
            substitution = dist[i - 1, j - 1] + 1 if differ else INFINITY
            copy = dist[i - 1, j - 1] if not differ else INFINITY
            best = min(insertion, deletion, substitution, copy)
            dist[i, j] = best
            if best == insertion:
                action[i, j] = action[i - 1, j]
            if best == deletion:
                action[i, j] = DELETION
            if best == substitution:
                action[i, j] = SUBSTITUTION
            if best == copy:
                action[i, j] = COPY
    return dist, action


def edit_distance(y, y_hat):
    """lvsr/error_rate.py:58-72"""
    return _edit_distance_matrix(y, y_hat)[0][-1, -1]


def wer(y, y_hat):
    """lvsr/error_rate.py:75-76 (the caller passes word or character sequences)."""
    return edit_distance(y, y_hat) / float(len(y))


# ---- alignment diagnostics (lvsr/expressions.py:4-25), numpy on host: weights (L,B,T'), masks (L,B) --------------------
def weights_std(weights, mask_outputs=None):
    w = numpy.asarray(weights, numpy.float64)
    pos = numpy.arange(w.shape[2])
    expected = (w * pos).sum(axis=2)
    expected2 = (w * pos ** 2).sum(axis=2)
    result = numpy.sqrt(numpy.maximum(expected2 - expected ** 2, 0.0))
    if mask_outputs is not None:
        result = result * mask_outputs
    return result.sum() / w.shape[0]


def monotonicity_penalty(weights, mask_x=None):
    c = numpy.cumsum(numpy.asarray(weights, numpy.float64), axis=2)
    pen = numpy.maximum(c[1:] - c[:-1], 0).sum(axis=2)
    if mask_x is not None:
        pen = pen * mask_x[1:]
    return pen.sum()


def entropy(weights, mask_x):
    w = numpy.asarray(weights, numpy.float64)
    ent = (w * numpy.log(w + 1e-7)).sum(axis=2)
    return (ent * mask_x).sum()
