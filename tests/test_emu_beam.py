"""Beam search through the emulated HIP generation step vs the hypotheses the reference produced (golden fixtures)."""
import numpy
import torch
import pytest
from numpy.testing import assert_allclose

from conftest import load_golden
from emu import emu_lib
from lvsr_amd import synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
from lvsr_amd.search import CandidateNotFoundError

CASES = ["tiny_conv_nowindow", "tiny_conv_median", "tiny_conv_logistic", "tiny_conv_relu", "tiny_conv_bottom", "tiny_conv_postmerge2", "tiny_content_embed",
         "tiny_content_relu", "tiny_conv_stack2", "tiny_content_stack3"]


def run_beam_case(case, device, lib):
    z, meta = load_golden(case)
    if not meta.get("beam"):
        pytest.skip("no beam fixture")
    params = synthetic.make_params(meta["cfg"], seed=meta["param_seed"], scale=meta["scale"])
    batch = synthetic.make_batch(meta["cfg"], meta["B"], meta["T"], meta["L"], seed=meta["batch_seed"], ragged=meta["ragged"])
    rec = SpeechRecognizer(device=device, params=params, lib=lib, net_config=meta["cfg"])
    for bi, b in enumerate(meta["beam"]):
        s = dict(b["settings"])
        utt = s.pop("utt", 0)
        bs = s.pop("beam_size")
        tl = int(batch["recordings_mask"][:, utt].sum())
        x = batch["recordings"][:tl, utt]
        rec.init_beam_search(bs)
        if b.get("error"):
            with pytest.raises(CandidateNotFoundError):
                rec.beam_search({"recordings": x}, **s)
            continue
        outs, costs = rec.beam_search({"recordings": x}, **s)
        assert outs == b["outputs"], (case, bi)                                  # bit-exact hypotheses
        assert_allclose(costs, b["costs"], rtol=2e-5, atol=2e-5)
        key = "analyze%d_cost" % bi
        if key in z.files and outs:
            c, w, _ = rec.analyze({"recordings": x}, numpy.array(outs[0]))
            assert_allclose(c, z[key], rtol=2e-4, atol=2e-5)
            assert_allclose(w, z["analyze%d_weights" % bi], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("case", CASES)
def test_beam_search_emulated(case):
    run_beam_case(case, "cpu", emu_lib())


def test_generation_states_are_not_overwritten_by_the_next_step():
    """The beam search keeps the tensors `generation_next_states` returned across the next `generation_logprobs` call (it
    only re-gathers them when hypotheses are dropped): they must not alias the step buffers."""
    z, meta = load_golden("tiny_conv_median")
    cfg = meta["cfg"]
    rec = SpeechRecognizer(device="cpu", params=synthetic.make_params(cfg, seed=5, scale=meta["scale"]), lib=emu_lib(), net_config=cfg)
    gen = rec.generator
    x = numpy.random.RandomState(0).normal(size=(24, cfg["input_dim"])).astype(numpy.float32)
    rec.compute_contexts(x)
    st = gen.generation_initial_states(3)
    S, W = st["states"], st["weights"]
    for step in range(3):
        gen.generation_logprobs(S, W, step)
        nxt = gen.generation_next_states(S, W, step, numpy.array([1, 2, 0]))
        S, W = nxt["states"], nxt["weights"]
        keepS, keepW = S.clone(), W.clone()
        gen.generation_logprobs(S, W, step + 1)
        assert torch.equal(S, keepS) and torch.equal(W, keepW)


# ---- the selection rule alone (lvsr_topk_smallest) against the reference's `_smallest` (search.py:221-242) -------------
def _reference_smallest(matrix, k):
    flat = matrix.ravel()
    k = min(k, flat.size)
    args = numpy.argpartition(flat, k)[:k] if flat.size > k else numpy.arange(flat.size)
    args = args[numpy.argsort(flat[args])]
    return numpy.unravel_index(args, matrix.shape), flat[args]


def check_smallest(bs):
    rng = numpy.random.RandomState(0)
    cases = [rng.normal(size=(5, 7)), rng.normal(size=(1, 33)), rng.normal(size=(16, 33)), rng.normal(size=(3, 2)),
             numpy.abs(rng.normal(size=(40, 33))) * 50, -numpy.abs(rng.normal(size=(9, 11)))]
    for m in cases:
        m = m.astype(numpy.float32)
        for k in (1, 4, 16, min(m.size, 200), min(m.size + 5, 256)):          # k <= 256 (kernel capacity)
            (r, c), v = bs._smallest(m, k)
            (rr, rc), rv = _reference_smallest(m, k)
            assert numpy.array_equal(v, rv) and numpy.array_equal(r, rr) and numpy.array_equal(c, rc)     # no ties: identical
    # adversarial ties: many equal values around (and inside) the cut.  The reference's order among EQUAL values is whatever
    # introselect + quicksort leave (not portable across numpy builds); the device rule is stable: equal values in flat-index
    # order.  What both must agree on is the multiset of values, and every device pick must be a smallest-possible index.
    for trial in range(20):
        m = rng.randint(0, 4, size=(rng.randint(1, 17), 33)).astype(numpy.float32) * 0.5
        if trial % 3 == 0:
            m[:, ::2] = -0.0                              # negative zero sorts before +0.0 only in the bit pattern
            m[:, 1::2] = 0.0
        k = int(rng.randint(1, 40))
        (r, c), v = bs._smallest(m, k)
        (_, _), rv = _reference_smallest(m, k)
        assert numpy.array_equal(numpy.sort(v), numpy.sort(rv)) if trial % 3 else numpy.allclose(v, rv)
        flat = r * m.shape[1] + c
        order = numpy.lexsort((flat, v)) if trial % 3 else numpy.arange(len(v))
        assert numpy.array_equal(order, numpy.arange(len(v)))                      # ascending values, then ascending index
        if trial % 3:
            kth = v[-1]
            ties_taken = flat[v == kth]
            all_ties = numpy.flatnonzero(m.ravel() == kth)
            assert numpy.array_equal(ties_taken, all_ties[: len(ties_taken)])     # the lowest indices of the tied value


def test_smallest_on_the_device_matches_the_reference_rule():
    from lvsr_amd.search import BeamSearch
    z, meta = load_golden("tiny_conv_median")
    rec = SpeechRecognizer(device="cpu", params=synthetic.make_params(meta["cfg"], seed=5), lib=emu_lib(), net_config=meta["cfg"])
    check_smallest(BeamSearch(4, rec))


def test_validate_solution_function_vetoes_finished_hypotheses():
    """search.py:372-374: a rejected hypothesis leaves the beam but is not recorded; with a validator the driver steps with the
    host in the loop and must give the same result as the free-running device loop when nothing is rejected."""
    z, meta = load_golden("tiny_conv_nowindow")
    params = synthetic.make_params(meta["cfg"], seed=meta["param_seed"], scale=meta["scale"])
    batch = synthetic.make_batch(meta["cfg"], meta["B"], meta["T"], meta["L"], seed=meta["batch_seed"], ragged=meta["ragged"])
    rec = SpeechRecognizer(device="cpu", params=params, lib=emu_lib(), net_config=meta["cfg"])
    b = meta["beam"][0]
    s = dict(b["settings"])
    s.pop("utt", 0)
    rec.init_beam_search(s.pop("beam_size"))
    x = batch["recordings"][: int(batch["recordings_mask"][:, 0].sum()), 0]
    seen = []
    outs, costs = rec.beam_search({"recordings": x}, validate_solution_function=lambda inp, toks: seen.append(list(toks)) or True, **s)
    assert outs == b["outputs"] and len(seen) >= len(outs)
    assert all(t[0] == meta["cfg"]["num_phonemes"] for t in seen)                 # the validator sees the initial pseudo-token too
    best = outs[0]
    outs2, _ = rec.beam_search({"recordings": x}, validate_solution_function=lambda inp, toks: list(toks[1:]) != best, **s)
    assert best not in outs2 and len(outs2) > 0                 # the search goes on without it (other hypotheses may appear)


def test_validator_sees_every_finished_hypothesis_once_the_patience_list_was_cut():
    """stop_on='patience' sorts the finished list and cuts it to beam_size BEFORE each position (search.py:306-309), so once more
    than beam_size hypotheses have finished the new ones land at slot beam_size, not behind the previous count: the validator must
    still see every one of them (search.py:372-374) and vetoed ones must not come back.  Compared with the oracle's host search
    under the same validator."""
    import torch
    from oracle import lvsr_oracle as O
    z, meta = load_golden("tiny_conv_nowindow")
    cfg = dict(meta["cfg"], max_decoded_length_scale=1.0)
    params = synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"])
    batch = synthetic.make_batch(cfg, meta["B"], meta["T"], meta["L"], seed=meta["batch_seed"], ragged=meta["ragged"])
    x = batch["recordings"][: int(batch["recordings_mask"][:, 0].sum()), 0]
    rec = SpeechRecognizer(device="cpu", params=params, lib=emu_lib(), net_config=cfg)
    rec.init_beam_search(2)
    kw = dict(char_discount=0.0, round_to_inf=1e9, stop_on="patience")
    seen = []

    def validate(inputs, tokens):               # veto every hypothesis with an even number of tokens
        seen.append([int(t) for t in tokens])
        return len(tokens) % 2 == 1
    outs, costs = rec.beam_search({"recordings": x}, validate_solution_function=validate, **kw)
    seen_dev, seen[:] = list(seen), []
    ref_outs, ref_costs = O.OracleRecognizer(cfg, params, dtype=torch.float32).beam_search(x, 2, validate_solution_function=validate, **kw)
    assert len(seen_dev) > 2, "the scenario needs more than beam_size finished hypotheses"
    assert seen_dev == seen                                       # the same hypotheses were shown to the validator, in the same order
    assert outs == ref_outs and all(len(o) % 2 == 0 for o in outs)          # (+1 initial pseudo-token = odd length as the validator counts)
    assert_allclose(costs, ref_costs, rtol=2e-5, atol=2e-5)


# ---- several utterances in one set of launches (BeamSearch.search_batch) ---------------------------------------------------------
@pytest.mark.parametrize("case", ["tiny_conv_median", "tiny_conv_nowindow", "tiny_content_embed", "tiny_conv_stack2",
                                  pytest.param("tiny_conv_logistic", marks=pytest.mark.slow),
                                  pytest.param("tiny_content_stack3", marks=pytest.mark.slow)])
def test_batched_search_equals_the_single_searches_emulated(case):
    """All utterances of the fixture's (ragged) batch decoded side by side — rows [g K, g K + K) of the state buffers belong to
    utterance g, windows / position counters / stopping rules / finished lists per utterance — give, utterance by utterance, the
    hypotheses of the single search (and, for the fixture's utterance, of the reference), under every setting of the fixture."""
    run_batched_case(case, "cpu", emu_lib())


def run_batched_case(case, device, lib, cost_tol=2e-5):
    z, meta = load_golden(case)
    params = synthetic.make_params(meta["cfg"], seed=meta["param_seed"], scale=meta["scale"])
    batch = synthetic.make_batch(meta["cfg"], meta["B"], meta["T"], meta["L"], seed=meta["batch_seed"], ragged=meta["ragged"])
    rec = SpeechRecognizer(device=device, params=params, lib=lib, net_config=meta["cfg"])
    lens = [int(batch["recordings_mask"][:, u].sum()) for u in range(meta["B"])]
    xs = [batch["recordings"][:tl, u] for u, tl in enumerate(lens)]
    assert len(set(lens)) > 1 or not meta["ragged"]
    seen = set()
    for b in meta["beam"]:
        s = dict(b["settings"])
        utt, bs = s.pop("utt", 0), s.pop("beam_size")
        key = repr(sorted(s.items())) + str(bs)
        rec.init_beam_search(bs)
        if key not in seen:
            seen.add(key)
            singles = []
            for x in xs:
                try:
                    singles.append(rec.beam_search({"recordings": x}, **s))
                except CandidateNotFoundError as e:
                    singles.append(e)
            batched = rec.beam_search_batch(xs, **s)
            assert len(batched) == len(xs)
            for u, (one, many) in enumerate(zip(singles, batched)):
                if isinstance(one, Exception):
                    assert type(many) is type(one), (case, u)
                    continue
                assert not isinstance(many, Exception), (case, u, many)
                assert many[0] == one[0], (case, u)
                assert_allclose(many[1], one[1], rtol=cost_tol, atol=cost_tol)
                # ... and bit for bit what the same kernels find for the utterance alone (a batch of one: the tiled readout merge like
                # every batched search; weighted averages added in the same order by the fused and the per-group kernel)
                if u < 2 or device != "cpu":          # (the emulator takes seconds per search: two utterances per setting there)
                    alone = rec.beam_search_batch([xs[u]], **s)[0]
                    assert alone[0] == many[0] and alone[1] == many[1], (case, u)
        if b.get("error"):
            assert isinstance(batched[utt], CandidateNotFoundError)
        else:
            assert batched[utt][0] == b["outputs"], (case, utt)
            assert_allclose(batched[utt][1], b["costs"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("case", ["tiny_conv_median", "tiny_conv_postmerge2",
                                  pytest.param("tiny_content_embed", marks=pytest.mark.slow),
                                  pytest.param("tiny_conv_stack2", marks=pytest.mark.slow)])
def test_batched_search_with_the_tiled_readout_merge_emulated(case, monkeypatch):
    """lvsr_readout_merge (the readout's merge products as 16-row MFMA tiles, used from 64 rows on) forced on for the small
    fixtures: same hypotheses as the single searches and the reference."""
    from lvsr_amd.bricks.generator import SequenceGenerator
    monkeypatch.setattr(SequenceGenerator, "MERGE_ROWS", 1)
    run_batched_case(case, "cpu", emu_lib(), cost_tol=1e-4)


def test_batched_search_edge_cases_emulated():
    """A batch of one (the single search), utterances of one and two frames next to long ones (attended length 1: every window
    clamps to it; a position limit of 1), and a beam wider than the number of candidates of the first position."""
    z, meta = load_golden("tiny_conv_median")
    params = synthetic.make_params(meta["cfg"], seed=meta["param_seed"], scale=meta["scale"])
    batch = synthetic.make_batch(meta["cfg"], meta["B"], meta["T"], meta["L"], seed=meta["batch_seed"], ragged=meta["ragged"])
    rec = SpeechRecognizer(device="cpu", params=params, lib=emu_lib(), net_config=meta["cfg"])
    x = batch["recordings"]
    xs = [x[:, 0], x[:1, 1], x[:2, 0], x[:7, 2]]
    for beam, kw in ((3, dict()), (12, dict(char_discount=0.3, stop_on="optimistic_future_cost"))):
        rec.init_beam_search(beam)
        singles = []
        for u in xs:
            try:
                singles.append(rec.beam_search({"recordings": u}, **kw))
            except CandidateNotFoundError as e:
                singles.append(e)
        one = rec.beam_search_batch(xs[:1], **kw)
        assert len(one) == 1 and one[0][0] == singles[0][0]
        many = rec.beam_search_batch(xs, **kw)
        for u, (a, b) in enumerate(zip(singles, many)):
            if isinstance(a, Exception):
                assert type(b) is type(a), u
            else:
                assert not isinstance(b, Exception), (u, b)
                assert b[0] == a[0], u
                assert_allclose(b[1], a[1], rtol=2e-5, atol=2e-5)
