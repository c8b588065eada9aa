"""configs[4] in miniature against the reference: beam 16, window_around_median(before 10, after 100), FST language model with
the settings of exp/wsj/decode.sh:12-25 on a mid-size network (T' = 150), hypotheses and costs as the reference's own
SpeechRecognizer(lm=...).beam_search produced them (tests/golden/mid_conv_lm_decode.npz, oracle/theano_harness/gen_golden.py
`mid_conv_lm_decode`).

What is compared: the ranked hypotheses up to length STABLE_LENGTH, token for token, and their costs.  The tail of the result
list — 80..95-character hypotheses that only finish because the search runs into max_length, costs 180..212 against 12..60 for
the real ones — is decided by float32 rounding accumulated over dozens of positions: from the 4th hypothesis of utterance 0 on (22 vs 27
characters, cost 50.3 vs 60.4) the float32 torch oracle and the HIP path agree with each other but not with the reference, the
float64 oracle agrees with the reference there but parts from it two entries later (oracle/lvsr_oracle.py beam_search on this
fixture) — no two implementations, the reference's own arithmetic in another precision included, agree on the tail."""
import math

import numpy
import pytest
from numpy.testing import assert_allclose

from conftest import load_golden
from lvsr_amd import lm as LM, synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
from lvsr_amd.search import CandidateNotFoundError


STABLE_LENGTH = 20

# configs[4] at FULL size (tests/golden/wsj_decode_full.npz, gen_golden.py `wsj_decode_full`): the WSJ-base network (4 x 256 BiGRU,
# D = 256, M = 512, 10 location filters of 201 taps), T = 800 -> T' = 200, beam 16, window_around_median(10, 100), FST LM, the
# exp/wsj/decode.sh settings, two utterances (32 and 2 finished hypotheses in the reference).  This is what puts the WSJ-size
# instantiations of the decode kernels (attdec_energy_kernel<10>, readout_step at P = 256, beam_select 16 x 33, the batch-1
# encoder at H = 256) against the reference instead of against themselves.  The head that float32 rounding leaves alone is
# shorter at this size: up to 13 characters the reference, the float32 / float64 oracles and the HIP path agree token for token
# and to 1e-4 in cost; the 18-character hypothesis already costs 37.796 (reference) / 37.738 (float32 oracle) / 37.767 (float64).
FULL_STABLE_LENGTH = 13


def _fst_from_arcs(arcs, V):
    f = LM.ArcFST(start=int(arcs[0][0]))
    for a, b, il, w in arcs:
        f.add_arc(int(a), int(b), int(il), float(w))
    f.isyms = dict([("<eps>", 0)] + [("c%d" % c, c + 1) for c in range(V)])
    return f, {"c%d" % c: c for c in range(V)}


def run_decode_case(device, lib, device_lm, utterances=None, fixture="mid_conv_lm_decode", stable=STABLE_LENGTH, min_checked=6):
    z, meta = load_golden(fixture)
    cfg = meta["cfg"]
    V = cfg["num_phonemes"]
    params = synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"], scales=meta.get("scales"))
    rec = SpeechRecognizer(device=device, params=params, lib=lib, net_config=cfg)
    fst, cmap = _fst_from_arcs(z["arcs"], V)
    kw = dict(nn_char_map=cmap, **meta["lm"])
    rec.set_language_model(LM.DeviceFSTLanguageModel(fst, device, lib=rec.lib, **kw) if device_lm else LM.FSTLanguageModel(fst, **kw))
    checked = 0
    for r in meta["beam"]:
        if utterances is not None and r["utt"] not in utterances:
            continue
        s = dict(r["settings"])
        rec.init_beam_search(s.pop("beam_size"))
        x = z["x%d" % r["utt"]]
        if r.get("error"):
            with pytest.raises(CandidateNotFoundError):
                rec.beam_search({"recordings": x}, **s)
            continue
        outs, costs = rec.beam_search({"recordings": x}, **s)
        if stable is None:          # the WHOLE ranked list: same number of hypotheses, every one token for token, every cost
            n = len(r["outputs"])
            assert len(outs) == n, "utterance %d: %d hypotheses, the reference has %d" % (r["utt"], len(outs), n)
        else:
            n = sum(1 for h in r["outputs"] if len(h) <= stable)
            assert n >= 1 and all(len(h) <= stable for h in r["outputs"][:n])
        assert outs[:n] == r["outputs"][:n], "utterance %d" % r["utt"]           # the ranked hypotheses, token for token
        assert_allclose(costs[:n], r["costs"][:n], rtol=1e-4, atol=1e-4)
        checked += n
    assert utterances is not None or checked >= min_checked
    return rec


@pytest.mark.gpu
@pytest.mark.parametrize("device_lm", [True, False])
def test_full_size_wsj_decode_matches_the_reference_gpu(gpu_device, device_lm):
    rec = run_decode_case(gpu_device, None, device_lm, fixture="wsj_decode_full", stable=FULL_STABLE_LENGTH, min_checked=8)
    if device_lm:
        run_decode_case(gpu_device, None, True, fixture="wsj_decode_full", stable=FULL_STABLE_LENGTH, min_checked=8)     # replayed step graph


# Round 4: the same decode on a WELL-CONDITIONED full-size fixture (tests/golden/wsj_decode_full2.npz, gen_golden.py
# `wsj_decode_full2`): per-group parameter scales (contractive recurrences, sharper energies: gen_golden.WSJ_COND_DECODE, found with
# tools/probes/wsj_conditioning_search.py) on which the reference, the float32 oracle and the float64 oracle agree on the WHOLE ranked
# list — 172 / 9 / 12 finished hypotheses of up to 117 characters for the three utterances, costs within 4e-6 relative.  Here the
# entire list is compared: its length, every hypothesis token for token, every cost to 1e-4 — north_star's "bit-exact indices".
@pytest.mark.gpu
@pytest.mark.parametrize("device_lm", [True, False])
def test_full_size_wsj_decode_whole_list_matches_the_reference_gpu(gpu_device, device_lm):
    rec = run_decode_case(gpu_device, None, device_lm, fixture="wsj_decode_full2", stable=None, min_checked=193)
    if device_lm:
        run_decode_case(gpu_device, None, True, fixture="wsj_decode_full2", stable=None, min_checked=193)     # replayed step graph
        assert rec._beam_search.last_stats["positions"] > 0


def run_batched_decode_case(device, lib, fixture, stable, min_checked, repeat=1):
    """The fixture's utterances decoded side by side (BeamSearch.search_batch, device language model) against the reference's
    hypotheses — `repeat` copies of the list, so that the batch is larger than the fixture and equal utterances must give equal results."""
    z, meta = load_golden(fixture)
    cfg = meta["cfg"]
    V = cfg["num_phonemes"]
    params = synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"], scales=meta.get("scales"))
    rec = SpeechRecognizer(device=device, params=params, lib=lib, net_config=cfg)
    fst, cmap = _fst_from_arcs(z["arcs"], V)
    rec.set_language_model(LM.DeviceFSTLanguageModel(fst, device, lib=rec.lib, nn_char_map=cmap, **meta["lm"]))
    rows = [r for r in meta["beam"]]
    settings = [dict(r["settings"]) for r in rows]
    assert all(s == settings[0] for s in settings), "the fixture's utterances share one set of decode settings"
    s = settings[0]
    rec.init_beam_search(s.pop("beam_size"))
    xs = [z["x%d" % r["utt"]] for r in rows] * repeat
    checked = 0
    for _ in range(2):                      # eager / captured, then the replayed step graph
        results = rec.beam_search_batch(xs, **s)
        assert len(results) == len(xs)
        for i, got in enumerate(results):
            r = rows[i % len(rows)]
            if r.get("error"):
                assert isinstance(got, CandidateNotFoundError)
                continue
            outs, costs = got
            if stable is None:
                n = len(r["outputs"])
                assert len(outs) == n, "utterance %d: %d hypotheses, the reference has %d" % (r["utt"], len(outs), n)
            else:
                n = sum(1 for h in r["outputs"] if len(h) <= stable)
            assert outs[:n] == r["outputs"][:n], "utterance %d" % r["utt"]
            assert_allclose(costs[:n], r["costs"][:n], rtol=1e-4, atol=1e-4)
            checked += n
    assert checked >= min_checked
    return rec


@pytest.mark.gpu
def test_full_size_wsj_decode_batched_whole_list_matches_the_reference_gpu(gpu_device):
    """configs[4] at full size, the well-conditioned fixture, 3 x 4 utterances side by side: the WHOLE ranked list of every one."""
    rec = run_batched_decode_case(gpu_device, None, "wsj_decode_full2", None, 2 * 4 * 193, repeat=4)
    assert rec._beam_search.last_stats["positions"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("beam", [5, 32])
def test_full_size_batched_decode_equals_single_searches_at_other_beam_widths_gpu(gpu_device, beam):
    """Row groups that are not the 16 rows of an MFMA tile (beam 5: a tile of 16 rows spans four searches; beam 32: two tiles per
    search, two passes of the weighted-average kernel) and utterances of different lengths (T' = 200 / 160 / 250: windows clamped to
    the utterance's own attended length, per-search position limits): every utterance of the batch == the same utterance decoded
    alone, whole ranked list, on the well-conditioned full-size fixture's network and language model."""
    z, meta = load_golden("wsj_decode_full2")
    cfg = meta["cfg"]
    params = synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"], scales=meta.get("scales"))
    rec = SpeechRecognizer(device=gpu_device, params=params, net_config=cfg)
    fst, cmap = _fst_from_arcs(z["arcs"], cfg["num_phonemes"])
    rec.set_language_model(LM.DeviceFSTLanguageModel(fst, gpu_device, lib=rec.lib, nn_char_map=cmap, **meta["lm"]))
    s = dict(meta["beam"][0]["settings"])
    s.pop("beam_size")
    rec.init_beam_search(beam)
    x0, x1, x3 = z["x0"], z["x1"], z["x3"]
    xs = [x0, x1[:640], numpy.concatenate([x3, x0[:200]], axis=0), x1]
    singles = [rec.beam_search({"recordings": x}, **s) for x in xs]
    for _ in range(2):
        batched = rec.beam_search_batch(xs, **s)
        for u, (one, many) in enumerate(zip(singles, batched)):
            assert not isinstance(many, Exception), (u, many)
            assert many[0] == one[0], "utterance %d: the ranked lists differ" % u
            assert_allclose(many[1], one[1], rtol=1e-4, atol=1e-4)


# Round 5: beam 200 — the width the reference's README recommends for its best numbers (exp/wsj/README.md:58-60, exp/wsj/decode.sh:12):
# 200 hypotheses x 33 characters = 6 600 candidates per position in lvsr_beam_select (capacity 8 192), row groups of 200 = 12.5 of the
# 16-row MFMA tiles of the merge / energy kernels, finished lists of up to 200 x (max_length + 1) entries.
def _full2_recognizer(device, lib=None):
    z, meta = load_golden("wsj_decode_full2")
    cfg = meta["cfg"]
    params = synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"], scales=meta.get("scales"))
    rec = SpeechRecognizer(device=device, params=params, lib=lib, net_config=cfg)
    fst, cmap = _fst_from_arcs(z["arcs"], cfg["num_phonemes"])
    rec.set_language_model(LM.DeviceFSTLanguageModel(fst, device, lib=rec.lib, nn_char_map=cmap, **meta["lm"]))
    s = dict(meta["beam"][0]["settings"])
    s.pop("beam_size")
    return z, meta, params, rec, s


# The REFERENCE at beam 200 (tests/golden/wsj_decode_beam200.npz, gen_golden.py `wsj_decode_beam200`: lvsr's SpeechRecognizer.beam_search
# on the first 400 frames of utterances 1 and 0 of the wsj_decode_full2 set, same network / language model / decode settings, beam_size
# 200 — 52 and 61 s of Theano's Python linker): 299 and 61 ranked hypotheses.  The float32 oracle reproduces both lists in order (costs
# within 5.6e-4: hypotheses of 100+ characters whose costs of ~40 are sums of as many float32 step costs).
def _beam200_reference():
    z, meta = load_golden("wsj_decode_beam200")
    rows = {r["utt"]: r for r in meta["beam"]}
    assert all(r["settings"]["beam_size"] == 200 for r in rows.values())
    return z, meta, rows


@pytest.mark.gpu
def test_beam_200_matches_the_reference_gpu(gpu_device):
    """Beam 200 on the MI355X against the reference's own beam-200 search: the WHOLE ranked list of both utterances (299 and 61
    hypotheses) token for token, costs to 1e-3 (and all but a few to 1e-4); eager / captured, then the replayed step graph."""
    z, meta, rows = _beam200_reference()
    _, _, params, rec, s = _full2_recognizer(gpu_device)
    rec.init_beam_search(200)
    for utt, r in rows.items():
        for rep in range(2):
            outs, costs = rec.beam_search({"recordings": z["x%d" % utt]}, **s)
            assert len(outs) == len(r["outputs"]), (utt, len(outs), len(r["outputs"]))
            assert outs == r["outputs"], "utterance %d" % utt
            assert_allclose(costs, r["costs"], rtol=1e-3, atol=1e-4)
            assert numpy.isclose(costs, r["costs"], rtol=1e-4, atol=1e-4).mean() > 0.95


@pytest.mark.gpu
def test_beam_200_batched_matches_the_reference_gpu(gpu_device):
    """The two reference utterances and a third, longer one side by side at beam 200 (600 rows): the reference's WHOLE lists, token for
    token, as for the single search (the batched search sums an utterance's weighted averages in the single search's order)."""
    z, meta, rows = _beam200_reference()
    z2, _, params, rec, s = _full2_recognizer(gpu_device)
    rec.init_beam_search(200)
    order = sorted(rows)
    xs = [z["x%d" % u] for u in order] + [z2["x3"][:480]]
    for rep in range(2):
        batched = rec.beam_search_batch(xs, **s)
        for u, many in zip(order, batched):
            assert not isinstance(many, Exception), (u, many)
            assert many[0] == rows[u]["outputs"], "utterance %d" % u
            assert_allclose(many[1], rows[u]["costs"], rtol=1e-3, atol=1e-4)


@pytest.mark.slow
def test_float32_oracle_reproduces_the_reference_at_beam_200():
    """oracle/lvsr_oracle.py beam_search at beam 200 against the same fixture (~90 s per utterance): both lists in order."""
    import torch
    from oracle import lvsr_oracle as O, lm_oracle as LO
    z, meta, rows = _beam200_reference()
    cfg = meta["cfg"]
    V = cfg["num_phonemes"]
    orc = O.OracleRecognizer(cfg, synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"], scales=meta["scales"]), dtype=torch.float32)
    arcs = [(int(a), int(b), int(il), float(w)) for a, b, il, w in z["arcs"]]
    lm = dict(dense=LO.DenseFST(arcs, arcs[0][0], V), remap={c: c + 1 for c in range(V)}, **meta["lm"])
    for utt, r in rows.items():
        s = dict(r["settings"])
        outs, costs = orc.beam_search(z["x%d" % utt], s.pop("beam_size"), lm=lm, **s)
        assert outs == r["outputs"]
        assert_allclose(costs, r["costs"], rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
def test_beam_200_batched_equals_single_searches_gpu(gpu_device):
    """Three utterances of different lengths side by side at beam 200 (600 rows: groups of 200 across the 16-row tiles, 6 600
    candidates per search in lvsr_beam_select) against each decoded alone, eager / captured and replayed: every hypothesis and every
    cost of every list, bit for bit — what a search finds does not depend on what shares its launches (round 6: attdec_group_wa_kernel
    adds an utterance's weighted averages in the order of the single search's fused kernel)."""
    z, meta, params, rec, s = _full2_recognizer(gpu_device)
    rec.init_beam_search(200)
    xs = [z["x1"][:400], z["x0"][:320], z["x3"][:480]]
    singles = [rec.beam_search({"recordings": x}, **s) for x in xs]
    assert all(len(o[0]) >= 100 for o in singles)
    for _ in range(2):
        batched = rec.beam_search_batch(xs, **s)
        for u, (one, many) in enumerate(zip(singles, batched)):
            assert not isinstance(many, Exception), (u, many)
            assert many[0] == one[0], "utterance %d: hypotheses" % u
            assert many[1] == one[1], "utterance %d: costs" % u
        if _ == 0:
            first = batched
        else:               # the replayed step graph == the captured one, bit for bit
            assert [m[0] for m in batched] == [m[0] for m in first] and [m[1] for m in batched] == [m[1] for m in first]


@pytest.mark.slow
def test_beam_200_emulated_equals_the_float32_oracle():
    """(~3 minutes of fiber emulation, hence `--runslow`.)  Beam 200 through the emulated kernels on a small network (20 characters: 4 000 candidates per position; 679 finished
    hypotheses): the whole ranked list of the float32 oracle, token for token."""
    import torch
    from emu import emu_lib
    from oracle import lvsr_oracle as O
    z, meta = load_golden("small_conv_median")
    cfg = meta["cfg"]
    params = synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"])
    batch = synthetic.make_batch(cfg, meta["B"], meta["T"], meta["L"], seed=meta["batch_seed"], ragged=meta["ragged"])
    x = batch["recordings"][: int(batch["recordings_mask"][:, 1].sum()), 1][:16]
    kw = dict(char_discount=0.2, round_to_inf=1e9, stop_on="optimistic_future_cost")
    ref = O.OracleRecognizer(cfg, params, dtype=torch.float32).beam_search(x, 200, **kw)
    rec = SpeechRecognizer(device="cpu", params=params, lib=emu_lib(), net_config=cfg)
    rec.init_beam_search(200)
    got = rec.beam_search({"recordings": x}, **kw)
    assert len(got[0]) >= 200 and got[0] == ref[0]
    assert_allclose(got[1], ref[1], rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_mid_size_decode_batched_matches_the_reference_gpu(gpu_device):
    run_batched_decode_case(gpu_device, None, "mid_conv_lm_decode", STABLE_LENGTH, 12, repeat=2)


@pytest.mark.slow
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_oracles_reproduce_the_whole_list_of_the_well_conditioned_decode(dtype):
    """Both precisions of the torch restatement against wsj_decode_full2: the whole ranked list (~1 minute per utterance)."""
    import torch
    from oracle import lvsr_oracle as O, lm_oracle as LO
    z, meta = load_golden("wsj_decode_full2")
    cfg = meta["cfg"]
    V = cfg["num_phonemes"]
    orc = O.OracleRecognizer(cfg, synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"], scales=meta["scales"]),
                             dtype=getattr(torch, dtype))
    arcs = [(int(a), int(b), int(il), float(w)) for a, b, il, w in z["arcs"]]
    lm = dict(dense=LO.DenseFST(arcs, arcs[0][0], V), remap={c: c + 1 for c in range(V)}, **meta["lm"])
    for r in meta["beam"]:
        s = dict(r["settings"])
        outs, costs = orc.beam_search(z["x%d" % r["utt"]], s.pop("beam_size"), lm=lm, **s)
        assert outs == r["outputs"]
        assert_allclose(costs, r["costs"], rtol=1e-5, atol=1e-5)


@pytest.mark.slow
def test_oracle_reproduces_the_full_size_wsj_decode():
    """The torch restatement (float32) against the same fixture: ~3 minutes per utterance, hence `--runslow`."""
    import torch
    from oracle import lvsr_oracle as O, lm_oracle as LO
    z, meta = load_golden("wsj_decode_full")
    cfg = meta["cfg"]
    V = cfg["num_phonemes"]
    orc = O.OracleRecognizer(cfg, synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"]), dtype=torch.float32)
    arcs = [(int(a), int(b), int(il), float(w)) for a, b, il, w in z["arcs"]]
    lm = dict(dense=LO.DenseFST(arcs, arcs[0][0], V), remap={c: c + 1 for c in range(V)}, **meta["lm"])
    checked = 0
    for r in meta["beam"]:
        s = dict(r["settings"])
        outs, costs = orc.beam_search(z["x%d" % r["utt"]], s.pop("beam_size"), lm=lm, **s)
        n = sum(1 for h in r["outputs"] if len(h) <= FULL_STABLE_LENGTH)
        assert outs[:n] == r["outputs"][:n]
        assert_allclose(costs[:n], r["costs"][:n], rtol=1e-4, atol=1e-4)
        checked += n
    assert checked >= 8


@pytest.mark.gpu
@pytest.mark.parametrize("device_lm", [True, False])
def test_mid_size_decode_matches_the_reference_gpu(gpu_device, device_lm):
    rec = run_decode_case(gpu_device, None, device_lm)
    if device_lm:
        run_decode_case(gpu_device, None, True)          # second time: the step graph is replayed from the first position on
        assert rec._beam_search.last_stats["positions"] > 0


@pytest.mark.slow
def test_mid_size_decode_matches_the_reference_emulated():
    """One utterance through the fiber emulator: ~6 minutes on the GPU-less container, hence `--runslow`."""
    from emu import emu_lib
    run_decode_case("cpu", emu_lib(), True, utterances=[1])
