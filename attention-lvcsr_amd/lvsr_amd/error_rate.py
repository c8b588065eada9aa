"""Edit distance / error rates for decode reports (reference: lvsr/error_rate.py:11-76).
Host-side integer DP: not a kernel candidate (one hypothesis pair per utterance)."""
import numpy

COPY, INSERTION, DELETION, SUBSTITUTION = 0, 1, 2, 3      # lvsr/error_rate.py:3-6
INFINITY = 10 ** 9


def _edit_distance_matrix(y, y_hat):
    """dist[i, j] = edit distance between y[:i] and y_hat[:j]; action[i, j] = last action of an optimal chain turning
    y_hat[:j] into y[:i], with the reference's tie rule (lvsr/error_rate.py:11-55 tests copy last, so it wins; then
    substitution, then deletion; a pure insertion inherits the action of the cell above).

    Row-at-a-time: the horizontal (deletion) dependency of a row is a running minimum, cur[j] = j + min_{k<=j}(t[k] - k),
    where t holds the best of the diagonal and vertical moves; the actions are read off the finished distance table."""
    a = numpy.asarray(list(y), dtype=object)
    b = numpy.asarray(list(y_hat), dtype=object)
    n, m = len(a), len(b)
    cols = numpy.arange(m + 1, dtype="int64")
    dist = numpy.empty((n + 1, m + 1), dtype="int64")
    action = numpy.zeros((n + 1, m + 1), dtype="int64")
    dist[0] = cols
    for i in range(1, n + 1):
        up = dist[i - 1]
        mismatch = (b != a[i - 1]).astype("int64") if m else numpy.zeros(0, dtype="int64")
        t = numpy.empty(m + 1, dtype="int64")
        t[0] = i
        t[1:] = numpy.minimum(up[:-1] + mismatch, up[1:] + 1)
        row = numpy.minimum.accumulate(t - cols) + cols
        dist[i] = row
        if m:
            diag_hit = row[1:] == up[:-1] + mismatch                       # copy (mismatch 0) or substitution (1)
            left_hit = row[1:] == row[:-1] + 1                             # deletion
            act = numpy.where(diag_hit, numpy.where(mismatch == 1, SUBSTITUTION, COPY),
                              numpy.where(left_hit, DELETION, action[i - 1, 1:]))
            action[i, 1:] = act
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
