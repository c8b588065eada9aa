"""Error-rate helpers against the reference's own known answers (tests/test_error_rate.py:9-27) and the decode driver on
the emulated recognizer."""
import numpy
from numpy.testing import assert_equal, assert_allclose

from lvsr_amd import error_rate as ER


def test_edit_distance_matrix_reference_golden():
    dist, action = ER._edit_distance_matrix("abdce", "abcd")
    assert_equal(dist, [[0, 1, 2, 3, 4], [1, 0, 1, 2, 3], [2, 1, 0, 1, 2], [3, 2, 1, 1, 1], [4, 3, 2, 1, 2], [5, 4, 3, 2, 2]])
    assert_equal(action, [[0, 0, 0, 0, 0], [0, 0, 2, 2, 2], [0, 0, 0, 2, 2], [0, 0, 0, 3, 0], [0, 0, 0, 0, 3], [0, 0, 0, 0, 3]])
    assert ER.edit_distance("abdce", "abcd") == 2
    assert_allclose(ER.wer("abdce", "abcd"), 0.4)
    assert ER.edit_distance([1, 2, 3], []) == 3 and ER.edit_distance([], []) == 0


def test_alignment_diagnostics():
    w = numpy.zeros((3, 1, 5))
    w[0, 0, 0] = w[1, 0, 2] = w[2, 0, 1] = 1.0                     # delta alignments: zero spread
    assert_allclose(ER.weights_std(w), 0.0, atol=1e-12)
    assert ER.monotonicity_penalty(w) > 0                            # step 2 moves backwards
    assert_allclose(ER.monotonicity_penalty(w[:2]), 0.0, atol=1e-12)
    assert_allclose(ER.entropy(w, numpy.ones((3, 1))), 3 * numpy.log(1 + 1e-7), atol=1e-9)


def _decode_driver(device, lib):
    from lvsr_amd import synthetic, decode
    from lvsr_amd.bricks.recognizer import SpeechRecognizer
    cfg = dict(input_dim=5, num_phonemes=6, dims_bidir=[4], dim_dec=5, dim_matcher=6, attention_type="content",
               post_merge_dims=None, embed_outputs=True, data_prepend_eos=False)
    rec = SpeechRecognizer(device=device, params=synthetic.make_params(cfg, seed=5), lib=lib, net_config=cfg)
    rng = numpy.random.RandomState(0)
    utts = [(rng.normal(size=(9, 5)).astype(numpy.float32), [1, 2, 5]), (rng.normal(size=(7, 5)).astype(numpy.float32), [3, 5])]
    rep = decode.search(rec, utts, beam_size=3, char_discount=0.3, to_words=lambda ls: ["w%d" % l for l in ls if l != 5])
    assert len(rep["per_utterance"]) == 2 and 0.0 <= rep["cer"] and "wer" in rep
    for row in rep["per_utterance"]:
        assert numpy.isfinite(row["groundtruth_cost"])
        assert row["char_errors"] == ER.edit_distance(row["groundtruth"], row["recognized"])
    # the same report with the utterances decoded side by side (decode.search(batch=))
    rep2 = decode.search(rec, utts, beam_size=3, char_discount=0.3, to_words=lambda ls: ["w%d" % l for l in ls if l != 5], batch=2)
    assert [r["recognized"] for r in rep2["per_utterance"]] == [r["recognized"] for r in rep["per_utterance"]]
    assert_allclose([r["search_cost"] for r in rep2["per_utterance"]], [r["search_cost"] for r in rep["per_utterance"]], rtol=1e-5)
    assert rep2["cer"] == rep["cer"] and rep2["wer"] == rep["wer"]
    return rep


def test_decode_driver_emulated():
    from emu import emu_lib
    _decode_driver("cpu", emu_lib())


import pytest        # noqa: E402


@pytest.mark.gpu
def test_decode_driver_gpu(gpu_device):
    """lvsr_amd.decode.search on the MI355X against the ORACLE (oracle/lvsr_oracle.py, pinned to the reference's goldens): the
    recognised sequence and its search cost = the oracle's host beam search on the same utterance, the ground-truth cost = the
    oracle's `analyze`, error counts from those."""
    import torch
    from oracle import lvsr_oracle as O
    from lvsr_amd import synthetic
    got = _decode_driver(gpu_device, None)
    cfg = dict(input_dim=5, num_phonemes=6, dims_bidir=[4], dim_dec=5, dim_matcher=6, attention_type="content",
               post_merge_dims=None, embed_outputs=True, data_prepend_eos=False)
    orc = O.OracleRecognizer(cfg, synthetic.make_params(cfg, seed=5), dtype=torch.float32)
    rng = numpy.random.RandomState(0)
    utts = [(rng.normal(size=(9, 5)).astype(numpy.float32), [1, 2, 5]), (rng.normal(size=(7, 5)).astype(numpy.float32), [3, 5])]
    tot_err = tot_len = 0
    for row, (x, gt) in zip(got["per_utterance"], utts):
        try:
            outs, costs = orc.beam_search(x, 3, char_discount=0.3, round_to_inf=1e9, stop_on="patience")
        except LookupError:               # no hypothesis ended (CandidateNotFoundError in the reference): the driver records an empty one
            outs, costs = [[]], [float("nan")]
            assert row.get("error") == "CandidateNotFoundError"
        assert row["recognized"] == outs[0]
        assert_allclose(row["search_cost"], costs[0], rtol=1e-4, atol=1e-5)
        gt_cost, _ = orc.analyze(x, numpy.asarray(gt))
        assert_allclose(row["groundtruth_cost"], float(gt_cost.sum()), rtol=1e-4)
        assert row["char_errors"] == ER.edit_distance(gt, outs[0])
        tot_err += row["char_errors"]
        tot_len += len(gt)
    assert_allclose(got["cer"], tot_err / float(tot_len))


def test_host_side_pieces_match_the_reference_functions():
    """tests/golden/misc_reference.npz (oracle/theano_harness/gen_misc_golden.py): the reference's own edit-distance tables,
    alignment diagnostics (Theano-evaluated) and initialisation schemes."""
    import json
    import numpy
    from conftest import golden_path
    from lvsr_amd import blocks_compat as BC
    z = numpy.load(golden_path("misc_reference"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    for k, p in enumerate(meta["pairs"]):
        dist, action = ER._edit_distance_matrix(p["y"], p["y_hat"])
        assert (dist == z["ed%d_dist" % k]).all() and (action == z["ed%d_action" % k]).all(), p
        if p["wer"] is not None:
            assert ER.wer(p["y"], p["y_hat"]) == p["wer"]
    w, m = z["expr_weights"], z["expr_mask"]
    assert_allclose(ER.weights_std(w), z["weights_std"], rtol=2e-4)
    assert_allclose(ER.weights_std(w, m), z["weights_std_masked"], rtol=2e-4)
    assert_allclose(ER.monotonicity_penalty(w), z["monotonicity_penalty"], rtol=1e-5)
    assert_allclose(ER.monotonicity_penalty(w, m), z["monotonicity_penalty_masked"], rtol=1e-5)
    assert_allclose(ER.entropy(w, m), z["entropy"], rtol=1e-5)
    schemes = dict(constant=BC.Constant(0.3), gauss=BC.IsotropicGaussian(0.1, 0.2), uniform_width=BC.Uniform(width=0.4),
                   uniform_std=BC.Uniform(mean=1.0, std=0.2), orth_square=BC.Orthogonal(), orth_wide=BC.Orthogonal(),
                   orth_tall=BC.Orthogonal())
    for name, shape in meta["inits"]:
        got = schemes[name].generate(numpy.random.RandomState(5), tuple(shape))
        ref = z["init_" + name]
        assert got.dtype == ref.dtype and got.shape == ref.shape
        assert_allclose(got, ref, rtol=1e-6, atol=1e-7, err_msg=name)
