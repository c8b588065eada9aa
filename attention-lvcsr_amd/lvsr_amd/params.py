"""Parameter storage: one flat fp32 buffer for values and one for gradients, named views in the
reference's Blocks naming.  A flat gradient buffer is what the data-parallel step
all-reduces (one RCCL all-reduce per step) and what the optimiser kernel walks.
"""
from collections import OrderedDict

import numpy
import torch

from .spec import parameter_shapes


class Workspace(object):
    """Named scratch buffers.  One flat allocation per name, grown geometrically and handed out as a view of the requested
    shape: memory is bounded by the largest shape ever asked for under a name (training sees a new (T, L) almost every
    minibatch), and the base pointer of a name stays put while its capacity suffices — captured hipGraphs bake pointers
    in and are keyed by them, so a stable pointer means replays keep hitting.  A buffer comes back zeroed only on its
    first allocation or with zero=True; callers must not rely on stale contents."""

    def __init__(self, device):
        self.device = device
        self._bufs = {}          # (name, dtype) -> flat tensor
        self.generation = 0      # bumped whenever a buffer is (re)allocated: pointers handed out earlier may be stale

    def get(self, name, shape, dtype=torch.float32, zero=False):
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        key = (name, dtype)
        flat = self._bufs.get(key)
        if flat is None or flat.numel() < n:
            cap = max(n, 1) if flat is None else max(n, int(flat.numel() * 1.5))
            flat = torch.zeros(cap, dtype=dtype, device=self.device)
            self.generation += 1
            self._bufs[key] = flat
            zero = False
        t = flat[:n].view(shape)
        if zero:
            t.zero_()
        return t

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self._bufs.values())


class ParameterStore(object):
    def __init__(self, cfg, device, values=None):
        self.shapes = parameter_shapes(cfg)
        self.device = torch.device(device)
        offs, total = OrderedDict(), 0
        for name, shape in self.shapes.items():
            n = int(numpy.prod(shape))
            offs[name] = (total, n)
            total += (n + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.offsets = offs
        self.version = 0          # bumped whenever parameter values change (packed copies key on it)
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        # the gradient bucket of data parallelism = [4 guard floats | gradients]: element 0 carries "a persistent cluster kernel of
        # this step gave up" (lvsr_guard_collect) through the all-reduce, so every rank skips such a step together (lvsr_opt_args.guard)
        self.grad_bucket = torch.zeros(total + 4, dtype=torch.float32, device=self.device)
        self.guard = self.grad_bucket[:1]
        self.grad = self.grad_bucket[4:]
        self.p = OrderedDict((k, self.flat[o:o + n].view(self.shapes[k])) for k, (o, n) in offs.items())
        self.g = OrderedDict((k, self.grad[o:o + n].view(self.shapes[k])) for k, (o, n) in offs.items())
        if values is not None:
            self.set_values(values)

    def num_parameters(self):
        return sum(n for _, n in self.offsets.values())

    def set_values(self, values):
        missing = set(self.shapes) - set(values)
        extra = set(values) - set(self.shapes)
        if missing or extra:
            raise ValueError("parameter names do not match: missing %s, unexpected %s" % (sorted(missing), sorted(extra)))
        for k, v in values.items():
            v = numpy.asarray(v, dtype=numpy.float32)
            if tuple(v.shape) != tuple(self.shapes[k]):
                raise ValueError("shape mismatch for %s: %s vs %s" % (k, v.shape, self.shapes[k]))
            self.p[k].copy_(torch.from_numpy(numpy.ascontiguousarray(v)))
        self.version += 1

    def get_values(self):
        return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in self.p.items())

    def get_grads(self):
        return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in self.g.items())
