"""CPU ORACLE of the reference's training step rules.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Plain numpy (float32) restatement; every function cites the reference lines it follows (paths relative to
/root/reference).  Parity status: PINNED by the reference's own known-answer tests
(libs/blocks/tests/algorithms/test_algorithms.py:96-119 momentum/adadelta, :182-192 step clipping, :199-225
variable clipping, :255-260 composite, :312-324 remove-not-finite) — see tests/test_optimizer.py.
"""
from collections import OrderedDict

import numpy

f32 = numpy.float32


def step_clipping(steps, threshold):
    """StepClipping.compute_steps, libs/blocks/blocks/algorithms/__init__.py:634-643."""
    if not threshold:
        return steps
    norm = numpy.sqrt(sum((numpy.asarray(s, f32).astype(f32) ** 2).sum(dtype=f32) for s in steps.values()), dtype=f32)
    mult = f32(1) if norm < f32(threshold) else f32(threshold) / norm
    return OrderedDict((k, numpy.asarray(s, f32) * mult) for k, s in steps.items())


class Momentum(object):
    """Momentum = Scale then BasicMomentum, :378-461."""
    def __init__(self, learning_rate=1.0, momentum=0.0):
        self.lr, self.mom, self.vel = f32(learning_rate), f32(momentum), {}

    def compute_steps(self, steps):
        out = OrderedDict()
        for k, s in steps.items():
            v = self.vel.get(k, numpy.zeros_like(s, dtype=f32))
            st = self.mom * v + self.lr * numpy.asarray(s, f32)
            self.vel[k] = st
            out[k] = st
        return out


class AdaDelta(object):
    """:464-515."""
    def __init__(self, decay_rate=0.95, epsilon=1e-6):
        if not 0.0 <= decay_rate <= 1.0:
            raise ValueError("decay rate needs to be in [0, 1]")
        self.rho, self.eps, self.ms_step, self.ms_dx = f32(decay_rate), f32(epsilon), {}, {}

    def compute_steps(self, steps):
        out = OrderedDict()
        for k, s in steps.items():
            s = numpy.asarray(s, f32)
            ms = self.rho * self.ms_step.get(k, numpy.zeros_like(s)) + (f32(1) - self.rho) * s * s
            dx = numpy.sqrt(self.ms_dx.get(k, numpy.zeros_like(s)) + self.eps) / numpy.sqrt(ms + self.eps) * s
            self.ms_step[k] = ms
            self.ms_dx[k] = self.rho * self.ms_dx.get(k, numpy.zeros_like(s)) + (f32(1) - self.rho) * dx * dx
            out[k] = dx
        return out


def variable_clipping(parameter, step, threshold, axis=None):
    """VariableClipping.compute_step, :701-720 (norm of the value AFTER the update)."""
    parameter, step = numpy.asarray(parameter, f32), numpy.asarray(step, f32)
    diff = parameter - step
    if axis is None:
        norms = numpy.sqrt((diff ** 2).sum(dtype=f32))
    else:
        norms = numpy.sqrt((diff ** 2).sum(axis=axis, keepdims=True, dtype=f32))
    with numpy.errstate(divide="ignore", invalid="ignore"):
        shrinking = parameter - (f32(threshold) / norms) * diff
    return numpy.where(norms > f32(threshold), shrinking, step).astype(f32)


def remove_not_finite(parameter, step, scaler=1):
    """RemoveNotFinite.compute_step, :855-861."""
    s = numpy.asarray(step, f32).sum(dtype=f32)
    if numpy.isnan(s) or numpy.isinf(s):
        return (f32(1) - f32(scaler)) * numpy.asarray(parameter, f32)
    return numpy.asarray(step, f32)


class AdaptiveClipping(object):
    """AdaptiveClipping.after_batch, lvsr/extensions.py:64-91 (Python floats; wired with decay_rate=0.998,
    burnin_period=500 in lvsr/main.py:616-619).  `threshold` is what StepClipping uses for the NEXT batch."""
    def __init__(self, initial_threshold, burnin_period=100, decay_rate=0.99):
        self.initial, self.burnin, self.decay = float(initial_threshold), burnin_period, decay_rate
        self.mean = self.mean2 = 0.0
        self.iterations_done = 0
        self.threshold = float(initial_threshold)

    def after_batch(self, total_gradient_norm):
        import math
        self.iterations_done += 1
        g = math.log(float(total_gradient_norm))
        self.mean = self.decay * self.mean + (1 - self.decay) * g
        self.mean2 = self.decay * self.mean2 + (1 - self.decay) * g ** 2
        std = max(self.mean2 - self.mean ** 2, 0.0) ** .5
        threshold = math.exp(self.mean + 1 * std)
        confidence = min(self.burnin, self.iterations_done) / float(self.burnin)
        threshold = confidence * threshold + (1 - confidence) * self.initial
        self.threshold = float(f32(min(threshold, 5 * self.initial)))            # shared floatX


def is_weight(name):
    """WEIGHT-role parameters (Linear.W, GRU state_to_state / state_to_gates, LookupTable.W); biases, initial
    states and conv1d.filters (allocated without a role, lvsr/bricks/attention.py:31-33) are not."""
    return name.endswith(".W") or name.endswith("state_to_state") or name.endswith("state_to_gates")


class TrainingRules(object):
    """CompositeRule([StepClipping] + [Momentum, AdaDelta] + [Restrict(VariableClipping)] + [RemoveNotFinite(0.0)])
    as assembled in lvsr/main.py:480-519, followed by `parameter -= step` (GradientDescent)."""
    def __init__(self, gradient_threshold=None, rules=("momentum",), scale=0.1, momentum=0.0, decay_rate=0.95,
                 epsilon=1e-8, max_norm=0.0, max_norm_exclude_lookup=False, nonfinite_scaler=0.0, burn_in_steps=0,
                 adaptive_clipping=None):
        self.thr = gradient_threshold
        self.burn = int(burn_in_steps or 0)                              # BurnIn, lvsr/algorithms.py:19-43
        self.adaptive = AdaptiveClipping(gradient_threshold, **adaptive_clipping) if adaptive_clipping else None
        self.core = []
        if "momentum" in rules:
            self.core.append(Momentum(scale, momentum))
        if "adadelta" in rules:
            self.core.append(AdaDelta(decay_rate, epsilon))
        self.max_norm, self.excl, self.scaler = max_norm, max_norm_exclude_lookup, nonfinite_scaler

    def step(self, params, grads):
        gl = OrderedDict((k, numpy.asarray(grads[k], f32)) for k in params)
        steps = step_clipping(gl, self.adaptive.threshold if self.adaptive else self.thr)
        if self.adaptive:       # total_gradient_norm = l2 norm of the raw gradients (GradientDescent, algorithms/__init__.py:232-236)
            self.adaptive.after_batch(numpy.sqrt(sum((g ** 2).sum(dtype=f32) for g in gl.values()), dtype=f32))
        burn = self.burn > 0
        self.burn = max(0, self.burn - 1)
        for rule in self.core:
            steps = rule.compute_steps(steps)
        new = OrderedDict()
        for k, p in params.items():
            s = steps[k]
            if self.max_norm and self.max_norm > 0 and is_weight(k) and not (self.excl and "lookuptable" in k):
                s = variable_clipping(p, s, self.max_norm, axis=0)
            s = remove_not_finite(p, s, self.scaler)
            if burn:
                s = s * f32(0)
            new[k] = (numpy.asarray(p, f32) - s).astype(f32)
        return new
