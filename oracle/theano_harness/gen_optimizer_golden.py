#!/usr/bin/env python3
"""Generate tests/golden/adaptive_clipping.npz from the REFERENCE's own classes, run from the scratch copy (make_scratch.py):
`lvsr.extensions.AdaptiveClipping.after_batch` driven with a sequence of gradient norms (its main loop replaced by the two
attributes it reads), wired exactly as lvsr/main.py:616-619 does (decay_rate=0.998, burnin_period=500) and once with a
short burn-in so the confidence ramp and the 5x cap are exercised; `lvsr.algorithms.BurnIn.compute_steps` evaluated by
Theano.  TEST INFRASTRUCTURE ONLY.

    oracle/theano_harness/run_gen.sh --script gen_optimizer_golden.py   (or: PYTHONPATH as in run_gen.sh, python3 this file)
"""
import json
import os
import sys
from collections import OrderedDict

import numpy
import theano
from theano import tensor

from blocks.algorithms import StepClipping
from lvsr.algorithms import BurnIn
from lvsr.extensions import AdaptiveClipping

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden", "adaptive_clipping.npz")


class FakeLog(object):
    current_row = {}


class FakeLoop(object):
    def __init__(self):
        self.log = FakeLog()
        self.status = {"iterations_done": 0}


def run_adaptive(norms, initial, **kw):
    clipping = StepClipping(initial)
    ext = AdaptiveClipping("total_gradient_norm", clipping, initial, **kw)
    ext.main_loop = FakeLoop()
    out = []
    for g in norms:
        ext.main_loop.status["iterations_done"] += 1          # MainLoop._run_iteration increments before after_batch
        ext.main_loop.log.current_row = {"total_gradient_norm": float(g)}
        ext.after_batch(None)
        out.append(float(clipping.threshold.get_value()))
    return numpy.array(out)


def run_burn_in(num_steps, n):
    rule = BurnIn(num_steps=num_steps)
    p = theano.shared(numpy.zeros(3, dtype="float32"))
    s = tensor.vector("s")
    steps, updates = rule.compute_steps(OrderedDict([(p, s)]))
    f = theano.function([s], list(steps.values()), updates=updates)
    x = numpy.array([1.0, -2.0, 3.0], dtype="float32")
    return numpy.array([f(x)[0] for _ in range(n)])


def main():
    rng = numpy.random.RandomState(3)
    norms = numpy.exp(rng.normal(1.0, 0.8, size=40)).astype("float32")
    norms[7] = 1e6                                             # an outlier: the 5x cap
    blob = dict(norms=norms,
                wired=run_adaptive(norms, 100.0, decay_rate=0.998, burnin_period=500),
                short=run_adaptive(norms, 2.0, decay_rate=0.9, burnin_period=5),
                burn_in=run_burn_in(3, 6))
    blob["meta"] = numpy.array(json.dumps(dict(wired=dict(initial=100.0, decay_rate=0.998, burnin_period=500),
                                               short=dict(initial=2.0, decay_rate=0.9, burnin_period=5),
                                               burn_in=dict(num_steps=3))))
    numpy.savez_compressed(OUT, **blob)
    print("wrote", OUT, blob["short"][:6], blob["burn_in"][:, 0])


if __name__ == "__main__":
    main()
