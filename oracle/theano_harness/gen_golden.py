#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own SpeechRecognizer (Theano, Python linker).

TEST INFRASTRUCTURE ONLY; runs in the build container:
    python oracle/theano_harness/make_scratch.py
    bash   oracle/theano_harness/run_gen.sh [case ...]

For every case: build lvsr.bricks.recognizer.SpeechRecognizer from the net config, check that the
reference's parameter names/shapes equal lvsr_amd.spec.parameter_shapes (checkpoint contract), load
lvsr_amd.synthetic.make_params into the shared variables, evaluate on lvsr_amd.synthetic.make_batch:
cost matrix (lvsr/bricks/recognizer.py:376-390), alignment weights / energies / encoder output (the
variables `analyze` extracts, recognizer.py:452-494) and d(sum cost)/d(parameters) (lvsr/main.py:340-345,
libs/blocks/blocks/algorithms/__init__.py:216-224); optionally beam search (recognizer.py:496-533) and
analyze.  Large cases store gradient fingerprints (norm, sum, probe-dot) instead of full gradients.
"""
import json
import os
import sys
import time

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))

import theano
from theano import tensor
from blocks.bricks import Rectifier, Maxout, Tanh
from blocks.bricks.recurrent import GatedRecurrent
from blocks.filter import VariableFilter
from blocks.roles import OUTPUT
from blocks.graph import ComputationGraph
from blocks.initialization import IsotropicGaussian, Constant
from blocks.model import Model
from lvsr.bricks.recognizer import SpeechRecognizer, SpeechBottom

from lvsr_amd import spec, synthetic

OUT = os.path.join(REPO, "tests", "golden")

ACT = {"maxout2": lambda: Maxout(2), "rectifier": Rectifier, "tanh": Tanh}


def build_reference(cfg, **extra):
    c = spec.normalize_net_config(cfg)
    kw = dict(
        input_dims={"recordings": c["input_dim"]}, input_num_chars={}, eos_label=c["eos_label"],
        num_phonemes=c["num_phonemes"], dim_dec=c["dim_dec"], dims_bidir=c["dims_bidir"],
        subsample=c["subsample"], enc_transition=GatedRecurrent, dec_transition=GatedRecurrent,
        use_states_for_readout=c["use_states_for_readout"], attention_type=c["attention_type"],
        conv_n=c["conv_n"], conv_num_filters=c["conv_num_filters"], dim_matcher=c["dim_matcher"],
        prior=dict(c["prior"]) if c["prior"] else None, criterion={"name": "log_likelihood"},
        energy_normalizer=c["energy_normalizer"] if c["attention_type"] == "content_and_conv" else None,
        bottom={"bottom_class": SpeechBottom, "activation": ACT[c["bottom_activation"]]() if c["bottom_dims"] else Rectifier(),
                "dims": list(c["bottom_dims"])},
        post_merge_dims=c["post_merge_dims"],
        post_merge_activation=ACT[c["post_merge_activation"]]() if c["post_merge_dims"] else None,
        embed_outputs=c["embed_outputs"], dim_output_embedding=c["dim_output_embedding"],
        data_prepend_eos=c["data_prepend_eos"], max_decoded_length_scale=c["max_decoded_length_scale"],
        dec_stack=c["dec_stack"], name="recognizer")
    kw.update(extra)
    rec = SpeechRecognizer(**kw)
    rec.weights_init = IsotropicGaussian(0.01)
    rec.biases_init = Constant(0.0)
    rec.push_initialization_config()
    rec.initialize()
    return rec


def run_case(name, cfg, B, T, L, ragged, param_seed, batch_seed, scale=1.0, store_full=True,
             beam=None, analyze=False, scales=None):
    t0 = time.time()
    rec = build_reference(cfg)
    cg = rec.get_cost_graph(batch=True)
    cost_matrix = cg.outputs[0]
    cost = cost_matrix.sum()
    params = Model(cost).get_parameter_dict()
    want = spec.parameter_shapes(cfg)
    got = {k: tuple(v.get_value().shape) for k, v in params.items()}
    assert set(got) == set(want), (sorted(set(got) ^ set(want)))
    for k in want:
        assert tuple(want[k]) == got[k], (k, want[k], got[k])
    values = synthetic.make_params(cfg, seed=param_seed, scale=scale, scales=scales)
    for k, v in params.items():
        v.set_value(values[k])
    batch = synthetic.make_batch(cfg, B, T, L, seed=batch_seed, ragged=ragged)

    weights, = VariableFilter(bricks=[rec.generator], name="weights")(cg)
    energies = VariableFilter(bricks=[rec.generator], name="energies")(cg)
    encoded, = VariableFilter(applications=[rec.encoder.apply], roles=[OUTPUT], name="encoded")(cg)
    names = list(want.keys())
    grads = tensor.grad(cost, [params[k] for k in names])
    outs = [cost_matrix, weights, encoded] + (energies[:1] if energies else []) + grads
    f = theano.function([rec.inputs["recordings"], rec.inputs_mask, rec.labels, rec.labels_mask], outs,
                        on_unused_input="warn")
    t1 = time.time()
    res = f(batch["recordings"], batch["recordings_mask"], batch["labels"], batch["labels_mask"])
    t2 = time.time()
    cm, w, enc = res[0], res[1], res[2]
    k = 3
    en = None
    if energies:
        en = res[3]
        k = 4
    g = dict(zip(names, res[k:]))
    meta = dict(name=name, cfg=cfg, B=B, T=T, L=L, ragged=bool(ragged), param_seed=param_seed,
                batch_seed=batch_seed, scale=scale, scales=scales, compile_s=t1 - t0, step_s=t2 - t1,
                theano_flags=os.environ.get("THEANO_FLAGS", ""))
    out = {"cost_matrix": cm, "weights_argmax": w.argmax(axis=2).astype(numpy.int64),
           "cost_sum": numpy.float64(cm.astype(numpy.float64).sum())}
    if store_full:
        out["weights"] = w
        out["encoded"] = enc
        if en is not None:
            out["energies"] = en
        for n in names:
            out["grad:" + n] = g[n]
    else:
        # full-size cases: keep the fixture small
        out["weights_sub"] = w[:, : min(B, 4)].astype(numpy.float32)
        out["encoded_fp"] = synthetic.fingerprint("encoded", enc)
        out["encoded_sub"] = enc[:: max(1, enc.shape[0] // 8), :2, :16].astype(numpy.float32)
        # round 5: the reference's gradient ELEMENTS at fixed sample positions (small tensors whole), so that the full-size
        # gradients are pinned element-wise to the reference itself and not only through three-number fingerprints
        for n in names:
            out["gsub:" + n] = g[n].ravel()[synthetic.grad_sample_index(n, g[n].shape)].astype(numpy.float32)
            out["gmax:" + n] = numpy.float32(numpy.abs(g[n]).max())
    out["grad_fp"] = numpy.stack([synthetic.fingerprint(n, g[n]) for n in names])
    out["grad_names"] = numpy.array(names)

    if beam:
        beams = []
        for bi, bs in enumerate(beam):
            rec.init_beam_search(bs["beam_size"])
            kw = {k2: v2 for k2, v2 in bs.items() if k2 not in ("beam_size", "utt")}
            b = bs.get("utt", 0)
            tl = int(batch["recordings_mask"][:, b].sum())
            x1 = batch["recordings"][:tl, b]
            try:
                outs_, costs_ = rec.beam_search({"recordings": x1}, **kw)
            except Exception as e:   # blocks.search.CandidateNotFoundError is part of the contract
                beams.append(dict(settings=bs, outputs=None, costs=None, error=type(e).__name__))
                continue
            beams.append(dict(settings=bs, outputs=[[int(t) for t in o] for o in outs_],
                              costs=[float(c_) for c_ in costs_]))
            if analyze and outs_:
                hyp = numpy.array(outs_[0], dtype=numpy.int64)
                a = rec.analyze({"recordings": x1}, hyp, hyp)
                out["analyze%d_cost" % bi] = a[0]
                out["analyze%d_weights" % bi] = a[1]
        meta["beam"] = beams
    out["meta"] = numpy.array(json.dumps(meta))
    os.makedirs(OUT, exist_ok=True)
    numpy.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("[golden] %s: compile %.1fs step %.1fs cost_sum %.6f" % (name, t1 - t0, t2 - t1, out["cost_sum"]))
    sys.stdout.flush()


def run_lm_case(name, cfg, T, param_seed, fst_seed, beams, lm_kwargs, scale=1.0, utterances=3, analyze_labels=0, scales=None,
                utt_ids=None):
    """Beam search WITH shallow fusion through the reference's own bricks (LanguageModel / FSTTransition / FSTCostsOp /
    ShallowFusionReadout, lvsr/bricks/language_models.py, lvsr/ops.py); the PyFST container is the harness stand-in
    (make_scratch.py `fst.py`).  The automaton travels in the fixture as an arc array."""
    import math
    import tempfile
    V = cfg["num_phonemes"]
    rng = numpy.random.RandomState(fst_seed)
    arcs, backoff = [], V + 1
    uni = rng.dirichlet(numpy.ones(V) * 2.0)
    for s_ in [0] + list(range(1, V + 1)):
        keep = rng.choice(V, size=max(2, V // 2), replace=False)
        pr = rng.dirichlet(numpy.ones(len(keep)))
        for c, pc in zip(keep, pr):
            arcs.append((s_, 1 + int(c), int(c) + 1, -math.log(0.8 * pc)))
        arcs.append((s_, backoff, 0, -math.log(0.2)))
    for c in range(V):
        arcs.append((backoff, 1 + c, c + 1, -math.log(uni[c])))
    path = os.path.join(tempfile.mkdtemp(), "lm.fst.txt")
    with open(path, "w") as fh:
        for (a, b, il, w) in arcs:
            fh.write("%d %d %d %d %r\n" % (a, b, il, il, float(w)))
    with open(path + ".isyms", "w") as fh:
        fh.write("<eps> 0\n")
        for c in range(V):
            fh.write("c%d %d\n" % (c, c + 1))
    cmap = {"c%d" % c: c for c in range(V)}
    rec = build_reference(cfg, lm=dict(path=path, **lm_kwargs), character_map=cmap)
    cg = rec.get_cost_graph(batch=True)
    params = Model(cg.outputs[0].sum()).get_parameter_dict()
    values = synthetic.make_params(cfg, seed=param_seed, scale=scale, scales=scales)
    for k, v in params.items():
        if k in values:
            v.set_value(values[k])
    missing = set(values) - set(params)
    assert not missing, missing
    out, results = {}, []
    for u in (utt_ids if utt_ids is not None else range(utterances)):
        t_u = time.time()
        x = numpy.random.RandomState(100 + u).normal(size=(T, cfg["input_dim"])).astype("float32")
        out["x%d" % u] = x
        for bs in beams:
            rec.init_beam_search(bs["beam_size"])
            kw = {k2: v2 for k2, v2 in bs.items() if k2 != "beam_size"}
            try:
                o, c = rec.beam_search({"recordings": x}, **kw)
                results.append(dict(utt=u, settings=bs, outputs=[[int(t) for t in h] for h in o], costs=[float(v) for v in c]))
            except Exception as e:
                results.append(dict(utt=u, settings=bs, outputs=None, costs=None, error=type(e).__name__))
        print("[golden] %s: utterance %d searched in %.0f s" % (name, u, time.time() - t_u))
        sys.stdout.flush()
    if analyze_labels:
        # SpeechRecognizer.analyze with the language model attached (recognizer.py:452-494 -> SequenceGenerator.evaluate with
        # `language_model.evaluate`, sequence_generators.py:286-296: the readout of every label is fused with the look-ahead costs of
        # the FST state reached by the PREVIOUS labels, the cost is LMEmitter's -readout[label]) — what the decode report of
        # lvsr/main.py:778-821 prints as groundtruth / recognized cost when `net.lm` is set
        rng2 = numpy.random.RandomState(7)
        for u in range(utterances):
            for j in range(analyze_labels):
                n = int(rng2.randint(2, 7))
                y = numpy.concatenate([rng2.randint(0, V - 1, size=n), [cfg.get("eos_label", V - 1) if cfg.get("eos_label") is not None else V - 1]]).astype("int64")
                a = rec.analyze({"recordings": out["x%d" % u]}, y, y)
                out["an_u%d_%d_labels" % (u, j)] = y
                out["an_u%d_%d_cost" % (u, j)] = numpy.asarray(a[0])
                out["an_u%d_%d_weights" % (u, j)] = numpy.asarray(a[1])
    out["arcs"] = numpy.array(arcs, dtype=numpy.float64)
    out["meta"] = numpy.array(json.dumps(dict(name=name, cfg=cfg, T=T, param_seed=param_seed, scale=scale, scales=scales, lm=lm_kwargs,
                                              beam=results, analyze_labels=int(analyze_labels))))
    numpy.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("[golden] %s: %s" % (name, [(r["utt"], r.get("error") or [len(h) for h in r["outputs"]][:3]) for r in results]))
    sys.stdout.flush()


def run_generate_case(name, cfg, B, T, n_steps, param_seed, batch_seed, scale=1.0):
    """SpeechRecognizer.generate (lvsr/bricks/recognizer.py:393-406 -> SequenceGenerator.generate,
    libs/blocks/blocks/bricks/sequence_generators.py:328-377) with its own Theano random stream.  The fixture records what
    the reference produced (outputs, costs, states, weights, weighted averages) and, per step and utterance, a uniform number
    that reproduces the emitted class under MultinomialFromUniform's rule (first class whose running float32 sum of
    probabilities exceeds the uniform): the midpoint of that class's interval, computed from the reference's own readout +
    softmax evaluated on the returned states / glimpses.  The MRG31k3p stream itself is not reproduced."""
    t0 = time.time()
    rec = build_reference(cfg)
    cgc = rec.get_cost_graph(batch=True)
    params = Model(cgc.outputs[0].sum()).get_parameter_dict()
    values = synthetic.make_params(cfg, seed=param_seed, scale=scale)
    for k, v in params.items():
        v.set_value(values[k])
    batch = synthetic.make_batch(cfg, B, T, 4, seed=batch_seed, ragged=True)
    generated = rec.get_generate_graph(use_mask=True, n_steps=n_steps)
    n_dec = spec.normalize_net_config(cfg)["dec_stack"]
    state_keys = ["states"] + ["states#%d" % l for l in range(1, n_dec)]      # RecurrentStack: one state sequence per layer
    keys = [k for k in state_keys + ["outputs", "weighted_averages", "weights", "costs"] if k in generated]
    cg = ComputationGraph([generated[k] for k in keys])
    f = theano.function([rec.inputs["recordings"], rec.inputs_mask], cg.outputs, updates=cg.updates)
    res = f(batch["recordings"], batch["recordings_mask"])
    out = dict(zip(keys, res))
    # the reference's readout on (previous states, current glimpses) -> class probabilities of every step
    gen = rec.generator
    sts = [tensor.tensor3("st%d" % l) for l in range(n_dec)]
    wa = tensor.tensor3("wa")
    fb = tensor.lmatrix("prev_outputs")
    readouts = gen.readout.readout(feedback=gen.readout.feedback(fb), weighted_averages=wa, **dict(zip(state_keys, sts)))
    probs_fn = theano.function([fb, wa] + sts, gen.readout.emitter.probs(readouts), on_unused_input="ignore")
    outputs = out["outputs"]
    dims = spec.Dims(cfg)
    prev_states = []
    for l, k in enumerate(state_keys):
        init_state = params[spec.decoder_layer_names(dims, l)["h0"]].get_value()
        prev_states.append(numpy.concatenate([numpy.tile(init_state[None, None, :], (1, B, 1)), out[k][:-1]], axis=0).astype("float32"))
    prev_outputs = numpy.concatenate([numpy.full((1, B), cfg["num_phonemes"], dtype="int64"), outputs[:-1]], axis=0)
    probs = probs_fn(prev_outputs, out["weighted_averages"].astype("float32"), *prev_states)
    if n_dec > 1:      # the fixture keeps the states of the layers side by side (the layout of lvsr_amd's stacked generator)
        out["states"] = numpy.concatenate([out.pop(k) if l else out[k] for l, k in enumerate(state_keys)], axis=2)
    uniforms = numpy.zeros(outputs.shape, dtype=numpy.float32)
    for t in range(outputs.shape[0]):
        for b in range(B):
            cum = numpy.float32(0)
            lo = numpy.float32(0)
            m = int(outputs[t, b])
            for c in range(probs.shape[2]):
                lo = cum
                cum = numpy.float32(cum + probs[t, b, c])
                if c == m:
                    break
            u = numpy.float32((numpy.float64(lo) + numpy.float64(cum)) / 2)
            # check: the rule applied to the reference's probabilities with this uniform returns the emitted class
            cc, pick = numpy.float32(0), 0
            for c in range(probs.shape[2]):
                cc = numpy.float32(cc + probs[t, b, c])
                if cc > u:
                    pick = c
                    break
            assert pick == m, (t, b, pick, m)
            uniforms[t, b] = u
    meta = dict(name=name, cfg=cfg, B=B, T=T, n_steps=n_steps, param_seed=param_seed, batch_seed=batch_seed, scale=scale,
                theano_flags=os.environ.get("THEANO_FLAGS", ""))
    save = {k: v for k, v in out.items()}
    save.update(uniforms=uniforms, probs=probs.astype(numpy.float32), meta=numpy.array(json.dumps(meta)))
    numpy.savez_compressed(os.path.join(OUT, name + ".npz"), **save)
    print("[golden] %s: %.1fs outputs %s cost sum %.5f" % (name, time.time() - t0, outputs[:, 0].tolist(), float(out["costs"].sum())))
    sys.stdout.flush()


def mid_cfg(prior, **kw):
    """A mid-size WSJ-like network for the decode fixture: 33 characters, one-hot feedback, maxout readout, 5 location filters."""
    cfg = dict(input_dim=40, num_phonemes=33, dims_bidir=[48, 48], subsample=[1, 2], dim_dec=64, dim_matcher=80,
               attention_type="content_and_conv", conv_n=40, conv_num_filters=5, prior=prior,
               post_merge_dims=[64], post_merge_activation="maxout2", embed_outputs=False,
               data_prepend_eos=False, max_decoded_length_scale=3.0)
    cfg.update(kw)
    return cfg


def tiny_cfg(prior, **kw):
    cfg = dict(input_dim=5, num_phonemes=6, dims_bidir=[3, 3], subsample=[1, 2], dim_dec=4, dim_matcher=7,
               attention_type="content_and_conv", conv_n=2, conv_num_filters=3, prior=prior,
               post_merge_dims=[8], post_merge_activation="maxout2", embed_outputs=False,
               data_prepend_eos=False)
    cfg.update(kw)
    return cfg


def small_cfg(prior, **kw):
    cfg = dict(input_dim=40, num_phonemes=20, dims_bidir=[32, 32], subsample=[1, 2], dim_dec=48, dim_matcher=64,
               attention_type="content_and_conv", conv_n=6, conv_num_filters=4, prior=prior,
               post_merge_dims=[32], post_merge_activation="maxout2", embed_outputs=False,
               data_prepend_eos=False)
    cfg.update(kw)
    return cfg


# Per-group parameter scales (substring of the parameter name -> factor, synthetic.make_params) of the well-conditioned full-size
# fixtures: contractive recurrences (random recurrent matrices at scale 1 amplify float32 rounding ~1.7x per label), sharp energies,
# a weak state -> energy coupling.  WSJ_COND_TRAIN was chosen for the GRADIENTS as well: with the first choice (energy_comp 3, no
# transform_states factor) costs and alignments of the two precisions agreed to 7e-8 but their gradients differed by up to 13 % of a
# tensor's maximum — the backward chain through 100 labels has a conditioning of its own; with this one 4.9e-4.
WSJ_COND_TRAIN = {"transition.state_to": 0.3, "gatedrecurrent.state_to": 0.5, "energy_comp": 2.0, "handler": 2.0, "transform_states": 0.3}
WSJ_COND_DECODE = {"transition.state_to": 0.15, "gatedrecurrent.state_to": 0.25, "energy_comp": 1.5, "transform_states": 0.5}

BEAMS = [dict(beam_size=4, char_discount=0.0, round_to_inf=1e9, stop_on="patience"),
         dict(beam_size=3, char_discount=0.3, round_to_inf=4.5, stop_on="optimistic_future_cost", utt=1),
         dict(beam_size=8, char_discount=0.1, round_to_inf=1e9, stop_on="optimistic_future_cost", utt=2)]

CASES = {
    "tiny_conv_expanding": lambda: run_case(
        "tiny_conv_expanding",
        tiny_cfg(dict(type="expanding", initial_begin=0, initial_end=3, min_speed=0.4, max_speed=1.3)),
        B=3, T=13, L=5, ragged=True, param_seed=1, batch_seed=11),
    "tiny_conv_nowindow": lambda: run_case(
        "tiny_conv_nowindow", tiny_cfg(None), B=3, T=13, L=5, ragged=True, param_seed=2, batch_seed=12,
        beam=BEAMS, analyze=True),
    "tiny_conv_median": lambda: run_case(
        "tiny_conv_median", tiny_cfg(dict(type="window_around_median", before=1, after=2)),
        B=3, T=13, L=5, ragged=True, param_seed=3, batch_seed=13, beam=BEAMS[:2], analyze=True),
    "tiny_conv_mean": lambda: run_case(
        "tiny_conv_mean", tiny_cfg(dict(type="window_around_mean", before=1.5, after=2.5)),
        B=3, T=13, L=5, ragged=True, param_seed=4, batch_seed=14),
    "tiny_conv_logistic": lambda: run_case(
        "tiny_conv_logistic", tiny_cfg(dict(type="window_around_mean", before=2, after=3), energy_normalizer="logistic"),
        B=3, T=13, L=5, ragged=True, param_seed=21, batch_seed=31, beam=BEAMS[:2], analyze=True),
    "tiny_conv_relu": lambda: run_case(
        "tiny_conv_relu", tiny_cfg(None, energy_normalizer="relu", embed_outputs=True),
        B=3, T=13, L=5, ragged=True, param_seed=23, batch_seed=32, beam=BEAMS[:1]),
    "tiny_conv_bottom": lambda: run_case(
        "tiny_conv_bottom", tiny_cfg(dict(type="window_around_median", before=2, after=2), bottom_dims=[6, 4],
                                     bottom_activation="rectifier"),
        B=3, T=13, L=5, ragged=True, param_seed=24, batch_seed=33, beam=BEAMS[:1]),
    # post_merge_dims with TWO entries (MLP([act, Identity], [8, 6, V]) behind Bias + act, recognizer.py:305-319), rectifier
    "tiny_conv_postmerge2": lambda: run_case(
        "tiny_conv_postmerge2", tiny_cfg(dict(type="window_around_median", before=2, after=2), post_merge_dims=[8, 6],
                                         post_merge_activation="rectifier"),
        B=3, T=13, L=5, ragged=True, param_seed=26, batch_seed=35, beam=BEAMS[:2], analyze=True),
    # dec_stack = 2 / 3: RecurrentStack decoder with skip connections (recognizer.py:250-262; wsj_jan_debug.yaml and three more
    # shipped WSJ configs): the attention and the readout see the states of every layer
    "tiny_conv_stack2": lambda: run_case(
        "tiny_conv_stack2", tiny_cfg(dict(type="window_around_median", before=2, after=2), dec_stack=2),
        B=3, T=13, L=5, ragged=True, param_seed=27, batch_seed=36, beam=BEAMS[:2], analyze=True),
    "tiny_content_stack3": lambda: run_case(
        "tiny_content_stack3",
        dict(input_dim=5, num_phonemes=6, dims_bidir=[4], dim_dec=5, dim_matcher=6, attention_type="content",
             post_merge_dims=None, embed_outputs=True, data_prepend_eos=False, dec_stack=3),
        B=3, T=9, L=4, ragged=True, param_seed=28, batch_seed=37, beam=BEAMS[:1]),
    "small_conv_stack2": lambda: run_case(
        "small_conv_stack2", small_cfg(dict(type="expanding", initial_begin=0, initial_end=4, min_speed=0.6, max_speed=2.2),
                                       dec_stack=2, embed_outputs=True, dim_output_embedding=10),
        B=5, T=50, L=12, ragged=True, param_seed=29, batch_seed=38,
        beam=[dict(beam_size=6, char_discount=0.2, round_to_inf=1e9, stop_on="optimistic_future_cost")]),
    "tiny_content_embed": lambda: run_case(
        "tiny_content_embed",
        dict(input_dim=5, num_phonemes=6, dims_bidir=[4], dim_dec=5, dim_matcher=6, attention_type="content",
             post_merge_dims=None, embed_outputs=True, data_prepend_eos=False),
        B=3, T=9, L=4, ragged=True, param_seed=5, batch_seed=15, beam=BEAMS[:2], analyze=True),
    "tiny_content_relu": lambda: run_case(
        "tiny_content_relu",
        dict(input_dim=5, num_phonemes=6, dims_bidir=[4, 3, 5], subsample=[2, 1, 3], dim_dec=5, dim_matcher=6,
             attention_type="content", post_merge_dims=[7], post_merge_activation="rectifier",
             embed_outputs=True, dim_output_embedding=3, data_prepend_eos=True),
        B=4, T=17, L=6, ragged=True, param_seed=6, batch_seed=16, beam=BEAMS[:1]),
    "small_conv": lambda: run_case(
        "small_conv", small_cfg(None), B=5, T=50, L=12, ragged=True, param_seed=7, batch_seed=17,
        beam=[dict(beam_size=6, char_discount=0.2, round_to_inf=1e9, stop_on="optimistic_future_cost")],
        analyze=True),
    "small_conv_median": lambda: run_case(
        "small_conv_median", small_cfg(dict(type="window_around_median", before=3, after=8)),
        B=5, T=50, L=12, ragged=True, param_seed=8, batch_seed=18,
        beam=[dict(beam_size=6, char_discount=0.2, round_to_inf=1e9, stop_on="optimistic_future_cost")]),
    "small_conv_expanding": lambda: run_case(
        # max_speed 2.2 is not a float32: step*float32(2.2) crosses integers differently from step*2.2 (k=5, 10)
        "small_conv_expanding",
        small_cfg(dict(type="expanding", initial_begin=0, initial_end=4, min_speed=0.6, max_speed=2.2)),
        B=5, T=50, L=12, ragged=True, param_seed=12, batch_seed=19),
    "tiny_conv_lm": lambda: run_lm_case(
        "tiny_conv_lm", tiny_cfg(dict(type="window_around_median", before=2, after=3), embed_outputs=True), T=14,
        param_seed=41, fst_seed=5, scale=6.0, utterances=4, lm_kwargs=dict(weight=0.5, no_transition_cost=20.0),
        beams=[dict(beam_size=4, char_discount=0.2, round_to_inf=1e9, stop_on="optimistic_future_cost"),
               dict(beam_size=3, char_discount=1.0, round_to_inf=15.0, stop_on="patience")]),
    # configs[4] in miniature: beam 16, window_around_median(before 10, after 100), FST LM with the settings of
    # exp/wsj/decode.sh:12-25 (lm.weight 0.5, no_transition_cost 20, char_discount 1.0, max length T/3), T' = 150
    "mid_conv_lm_decode": lambda: run_lm_case(
        "mid_conv_lm_decode", mid_cfg(dict(type="window_around_median", before=10, after=100)), T=300,
        # scale 2.0: peaked alignments, and still well conditioned (at 4.0 the float32 and float64 oracles part ways: saturated
        # tanh units make energies of distant positions tie, and rounding picks the winner)
        param_seed=51, fst_seed=9, scale=2.0, utterances=3, lm_kwargs=dict(weight=0.5, no_transition_cost=20.0),
        beams=[dict(beam_size=16, char_discount=1.0, round_to_inf=1e9, stop_on="optimistic_future_cost")]),
    # the same network, teacher forced (cost matrix, alignments, gradients): long attended sequence (T' = 150), wide location
    # filters (81 taps), 5 filters, window of 110 positions — none of which the tiny / small cases reach
    "mid_conv_median": lambda: run_case(
        "mid_conv_median", mid_cfg(dict(type="window_around_median", before=10, after=100)),
        B=3, T=300, L=20, ragged=True, param_seed=51, batch_seed=52, scale=2.0),
    # configs[4] at FULL size: the WSJ-base network (4x256 BiGRU, D = 256, M = 512, 10 filters of 201 taps, T = 800 -> T' = 200),
    # beam 16, window_around_median(10, 100), FST LM, exp/wsj/decode.sh:12-25 settings.  Scale 2.0 chosen with
    # tools/probes/wsj_decode_conditioning.py: the float32 and float64 oracles agree on the first 6-7 ranked hypotheses (up to 18
    # characters) there; at scale 1.0 only two hypotheses ever finish
    "wsj_decode_full": lambda: run_lm_case(
        "wsj_decode_full", dict(spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100)),
                                max_decoded_length_scale=3.0), T=800,
        param_seed=10, fst_seed=9, scale=2.0, utterances=2, lm_kwargs=dict(weight=0.5, no_transition_cost=20.0),
        beams=[dict(beam_size=16, char_discount=1.0, round_to_inf=1e9, stop_on="optimistic_future_cost")]),
    # round 4: the same at parameter scales on which the float32 and the float64 oracle agree on the WHOLE ranked list (WSJ_COND_DECODE,
    # tools/probes/wsj_conditioning_search.py decode: utterances 0 / 1 / 3 -> 172 / 9 / 12 finished hypotheses, costs within 2e-4)
    "wsj_decode_full2": lambda: run_lm_case(
        "wsj_decode_full2", dict(spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100)),
                                 max_decoded_length_scale=3.0), T=800,
        param_seed=10, fst_seed=9, scale=2.0, scales=WSJ_COND_DECODE, utt_ids=[0, 1, 3],
        lm_kwargs=dict(weight=0.5, no_transition_cost=20.0),
        beams=[dict(beam_size=16, char_discount=1.0, round_to_inf=1e9, stop_on="optimistic_future_cost")]),
    # round 5: beam 200 — the width exp/wsj/README.md:58-60 recommends — on the same network and language model, 400-frame utterances
    # (the first 400 frames of the wsj_decode_full2 utterances: RandomState(100 + u) is drawn row by row), T' = 100, up to 133 positions
    "wsj_decode_beam200": lambda: run_lm_case(
        "wsj_decode_beam200", dict(spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100)),
                                   max_decoded_length_scale=3.0), T=400,
        param_seed=10, fst_seed=9, scale=2.0, scales=WSJ_COND_DECODE, utt_ids=[1, 0],
        lm_kwargs=dict(weight=0.5, no_transition_cost=20.0),
        beams=[dict(beam_size=200, char_discount=1.0, round_to_inf=1e9, stop_on="optimistic_future_cost")]),
    # cost / analyze WITH the language model (both weightings of the fusion that the shipped decode scripts use)
    "tiny_conv_lm_analyze": lambda: run_lm_case(
        "tiny_conv_lm_analyze", tiny_cfg(dict(type="window_around_median", before=2, after=3), embed_outputs=True), T=14,
        param_seed=41, fst_seed=5, scale=6.0, utterances=2, lm_kwargs=dict(weight=0.5, no_transition_cost=20.0),
        beams=[dict(beam_size=3, char_discount=0.2, round_to_inf=1e9, stop_on="optimistic_future_cost")], analyze_labels=3),
    "tiny_conv_lm_analyze_tot": lambda: run_lm_case(
        "tiny_conv_lm_analyze_tot", tiny_cfg(None, embed_outputs=False), T=12,
        param_seed=43, fst_seed=6, scale=4.0, utterances=2,
        lm_kwargs=dict(weight=0.8, no_transition_cost=15.0, normalize_am_weights=False, normalize_lm_weights=True, normalize_tot_weights=True, am_beta=0.7),
        beams=[dict(beam_size=3, char_discount=0.0, round_to_inf=1e9, stop_on="patience")], analyze_labels=3),
    "tiny_conv_generate": lambda: run_generate_case(
        "tiny_conv_generate", tiny_cfg(dict(type="window_around_median", before=1, after=2)), B=3, T=13, n_steps=7,
        param_seed=3, batch_seed=13, scale=2.0),
    "tiny_conv_stack2_generate": lambda: run_generate_case(
        "tiny_conv_stack2_generate", tiny_cfg(dict(type="window_around_median", before=1, after=2), dec_stack=2), B=3, T=13, n_steps=7,
        param_seed=27, batch_seed=36, scale=2.0),
    "small_conv_generate": lambda: run_generate_case(
        "small_conv_generate", small_cfg(None), B=4, T=40, n_steps=12, param_seed=7, batch_seed=17, scale=2.0),
    "timit_tiny": lambda: run_case(
        "timit_tiny", spec.timit_tiny(), B=2, T=200, L=40, ragged=False, param_seed=9, batch_seed=1234,
        store_full=False),
    "wsj_base": lambda: run_case(
        "wsj_base", spec.wsj_base(), B=16, T=800, L=100, ragged=False, param_seed=10, batch_seed=1234,
        store_full=False),
    # round 4: WSJ-base under the prior the shipped models train with, window_around_median (wsj_paper.yaml:7-10,
    # lvsr/bricks/attention.py:138-157), at parameter scales on which the float32 and the float64 oracle agree on all 100 x 16 alignment
    # argmax / window centres, to 1e-9 on the summed cost and to 5e-4 of a tensor's maximum on every gradient (WSJ_COND_TRAIN,
    # tools/probes/wsj_conditioning_search.py train); the
    # alignment of this seed travels through the utterance instead of sticking to its end
    "wsj_base_median": lambda: run_case(
        "wsj_base_median", spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100)), B=16, T=800, L=100,
        ragged=False, param_seed=13, batch_seed=1234, scales=WSJ_COND_TRAIN, store_full=False),
    # round 5: configs[1] RAGGED (T_i in [400, 800], L_i in [50, 100]; utterance 0 full length): the reference's mask semantics at
    # full size — the backward direction starting in padding (recurrent.py:617-619), masked label steps (attention.py:650-662),
    # fuel Padding (transformers/__init__.py:691-720) — through the 8/16-work-group cluster kernels
    "wsj_base_ragged": lambda: run_case(
        "wsj_base_ragged", spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100)), B=16, T=800, L=100,
        ragged=True, param_seed=13, batch_seed=1234, scales=WSJ_COND_TRAIN, store_full=False),
    # round 5: configs[1] under window_around_mean (lvsr/bricks/attention.py:135-137,148-157), window 30 before / 40 after
    "wsj_base_mean": lambda: run_case(
        "wsj_base_mean", spec.wsj_base(prior=dict(type="window_around_mean", before=30, after=40)), B=16, T=800, L=100,
        ragged=False, param_seed=13, batch_seed=1234, scales=WSJ_COND_TRAIN, store_full=False),
    # the WSJ-base network with the two-layer RecurrentStack decoder of wsj_jan_wsj13v2.yaml, full size (fingerprints).  Scale 0.7:
    # at 1.0 this seed's two-layer recurrence amplifies float32 rounding along the 100 labels (float32 and float64 oracles 4.7e-3
    # apart in the costs, 4e-2 in the alignments); at 0.7 they agree to 7e-7 / 2e-7 with alignments that are still peaked (max 0.14)
    "wsj_stack2": lambda: run_case(
        "wsj_stack2", dict(spec.wsj_base(), dec_stack=2), B=16, T=800, L=100, ragged=False, param_seed=13, batch_seed=1234,
        scale=0.7, store_full=False),
    # the README-recommended model at full size (exp/wsj/configs/wsj_paper7.yaml chain: 4 x 250 BiGRU on 123 features, D = M = 250,
    # one location filter, rectifier post-merge, embedded feedback, batch 10) with the expanding prior of its pre-training stage
    "wsj_paper": lambda: run_case(
        "wsj_paper", dict(spec.wsj_paper(), prior=dict(type="expanding", initial_begin=0, initial_end=40, min_speed=1.2, max_speed=2.2)),
        B=10, T=800, L=100, ragged=False, param_seed=15, batch_seed=1234, store_full=False),
    "wsj_deep": lambda: run_case(
        "wsj_deep", spec.wsj_deep(), B=8, T=1500, L=190, ragged=False, param_seed=11, batch_seed=1234,
        store_full=False),
}

if __name__ == "__main__":
    which = sys.argv[1:] or [k for k in CASES if k not in ("wsj_base", "wsj_deep", "wsj_stack2", "wsj_paper", "mid_conv_lm_decode", "wsj_decode_full",
                                                          "wsj_base_median", "wsj_decode_full2", "wsj_base_ragged", "wsj_base_mean", "wsj_decode_beam200")]
    for k in which:
        CASES[k]()
