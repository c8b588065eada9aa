#!/usr/bin/env python3
"""Generate tests/golden/misc_reference.npz from the reference's own small host-side pieces, run from the scratch copy:
  * lvsr.error_rate._edit_distance_matrix / wer on random sequence pairs over a small alphabet (many ties);
  * lvsr.expressions.weights_std / monotonicity_penalty / entropy evaluated by Theano on random alignments;
  * blocks.initialization.{Constant, IsotropicGaussian, Uniform, Orthogonal}.generate with seeded RandomStates.
TEST INFRASTRUCTURE ONLY."""
import json
import os

import numpy
import theano
from theano import tensor

from blocks import initialization as BI
from lvsr import error_rate as ER
from lvsr import expressions as EX

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden", "misc_reference.npz")


def main():
    rng = numpy.random.RandomState(11)
    blob, pairs = {}, []
    for k in range(40):
        y = rng.randint(0, 3, size=rng.randint(0, 9)).tolist()
        h = rng.randint(0, 3, size=rng.randint(0, 9)).tolist()
        dist, action = ER._edit_distance_matrix(y, h)
        blob["ed%d_dist" % k], blob["ed%d_action" % k] = dist, action
        pairs.append(dict(y=y, y_hat=h, wer=(ER.wer(y, h) if y else None)))
    L, B, T = 6, 3, 9
    w = rng.dirichlet(numpy.ones(T), size=(L, B)).astype("float32")
    mo = (rng.rand(L, B) > 0.3).astype("float32")
    W, M = tensor.tensor3("w"), tensor.matrix("m")
    blob["expr_weights"], blob["expr_mask"] = w, mo
    blob["weights_std"] = theano.function([W], EX.weights_std(W))(w)
    blob["weights_std_masked"] = theano.function([W, M], EX.weights_std(W, M))(w, mo)
    blob["monotonicity_penalty"] = theano.function([W], EX.monotonicity_penalty(W))(w)
    blob["monotonicity_penalty_masked"] = theano.function([W, M], EX.monotonicity_penalty(W, M))(w, mo)
    blob["entropy"] = theano.function([W, M], EX.entropy(W, M))(w, mo)
    inits = [("constant", BI.Constant(0.3), (4, 3)), ("gauss", BI.IsotropicGaussian(0.1, 0.2), (5, 7)),
             ("uniform_width", BI.Uniform(width=0.4), (6, 2)), ("uniform_std", BI.Uniform(mean=1.0, std=0.2), (3, 3)),
             ("orth_square", BI.Orthogonal(), (6, 6)), ("orth_wide", BI.Orthogonal(), (4, 7)), ("orth_tall", BI.Orthogonal(), (7, 4))]
    for name, scheme, shape in inits:
        blob["init_" + name] = scheme.generate(numpy.random.RandomState(5), shape)
    blob["meta"] = numpy.array(json.dumps(dict(pairs=pairs, inits=[[n, list(s)] for n, _, s in inits])))
    numpy.savez_compressed(OUT, **blob)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
