"""Deterministic synthetic parameters and batches (numpy only).

There is no dataset or checkpoint access, so the benchmark, the parity tests and the golden-vector
generator all draw from here.  Batch layout is the reference's: time-major
``recordings (T,B,F) f32``, ``recordings_mask (T,B) f32``, ``labels (L,B) i64``, ``labels_mask (L,B) f32``,
zero padded with mask 1 on real positions (fuel Padding semantics), `<eol>` appended to every label
sequence (lvsr/datasets/__init__.py:267-270).
"""
import zlib
import numpy

from .spec import parameter_shapes, Dims


def _rng_for(name, seed):
    return numpy.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))


def make_params(cfg, seed=1, scale=1.0, scales=None):
    """Parameters by reference name.  Fan-in scaled Gaussians so activations are O(1) and attention /
    gates are far from their trivial fixed points (a near-zero init would hide indexing bugs).

    `scale` multiplies every weight matrix; biases and initial states ~ N(0, 0.3) / N(0, 0.2).
    `scales`: optional {substring of the parameter name: factor}, applied on top of `scale` to every parameter (biases and
    initial states too) whose name contains the substring — the per-group conditioning of the full-size fixtures
    (tools/probes/wsj_conditioning_search.py); several matching substrings multiply.
    """
    out = {}
    for name, shape in parameter_shapes(cfg).items():
        rng = _rng_for(name, seed)
        if name.endswith("initial_state"):
            v = rng.normal(0.0, 0.2, shape)
        elif name.endswith(".b"):
            v = rng.normal(0.0, 0.3, shape)
        elif name.endswith("conv1d.filters"):
            v = rng.normal(0.0, 1.0, shape) * scale
        elif name.endswith("energy_comp/linear.W"):
            v = rng.normal(0.0, 2.0 / numpy.sqrt(shape[0]), shape) * scale * 3.0
        elif name.endswith("lookuptable.W"):
            v = rng.normal(0.0, 1.0, shape) * scale
        elif name.split("#")[0].split(".")[0].endswith(("fork/fork_inputs", "fork/fork_gate_inputs")) and name.endswith(".W"):
            # generator fork of a one-hot feedback has fan-in 1 effectively
            fan = 1.0 if "/generator/fork/" in name else shape[0]
            v = rng.normal(0.0, 1.0 / numpy.sqrt(fan), shape) * scale
        else:
            v = rng.normal(0.0, 1.0 / numpy.sqrt(shape[0]), shape) * scale
        for key, factor in (scales or {}).items():
            if key in name:
                v = v * factor
        out[name] = numpy.ascontiguousarray(v, dtype=numpy.float32)
    return out


def make_batch(cfg, B, T, L, seed=1234, ragged=False):
    """One synthetic minibatch: features ~ N(0,1) (global-CMVN'd fbank look-alike), labels uniform in
    [0, V-2] with <eol>=eos_label appended.  ragged=True draws T_i ~ U{T/2..T}, L_i ~ U{L/2..L}."""
    d = Dims(cfg)
    rng = numpy.random.RandomState(seed)
    x = rng.normal(0.0, 1.0, (T, B, d.F)).astype(numpy.float32)
    if ragged:
        t_len = rng.randint(max(1, T // 2), T + 1, size=B)
        l_len = rng.randint(min(L, max(2, L // 2)), L + 1, size=B)
        t_len[0] = T
        l_len[0] = L
    else:
        t_len = numpy.full(B, T)
        l_len = numpy.full(B, L)
    eos = d.cfg["eos_label"]
    labels = rng.randint(0, max(1, d.V - 1), size=(L, B)).astype(numpy.int64)
    x_mask = numpy.zeros((T, B), numpy.float32)
    y_mask = numpy.zeros((L, B), numpy.float32)
    for b in range(B):
        x_mask[: t_len[b], b] = 1.0
        x[t_len[b]:, b] = 0.0
        y_mask[: l_len[b], b] = 1.0
        labels[l_len[b] - 1, b] = eos
        labels[l_len[b]:, b] = 0
    return dict(recordings=x, recordings_mask=x_mask, labels=labels, labels_mask=y_mask)


def shard_batch(batch, rank, world):
    """Utterance sharding for data parallelism: rank r takes utterances r::world."""
    return {k: numpy.ascontiguousarray(v[:, rank::world]) for k, v in batch.items()}


def grad_probe(name, shape):
    """Fixed random direction used to fingerprint a large gradient tensor in the golden fixtures."""
    return _rng_for(name + "#probe", 0).normal(0.0, 1.0, shape).astype(numpy.float64)


def fingerprint(name, g):
    """(l2 norm, sum, dot with the fixed probe) of a tensor, in float64."""
    g = numpy.asarray(g, numpy.float64)
    return numpy.array([numpy.sqrt((g * g).sum()), g.sum(), (g * grad_probe(name, g.shape)).sum()])


def grad_sample_index(name, shape, n=2048):
    """Fixed flat indices (sorted, without repetition) at which the full-size golden fixtures keep the reference's gradient
    ELEMENTS themselves (`gsub:<name>`), next to the three-number fingerprint: small tensors whole, large ones n elements."""
    size = int(numpy.prod(shape))
    if size <= n:
        return numpy.arange(size)
    return numpy.sort(_rng_for(name + "#sample", 0).choice(size, size=n, replace=False))
