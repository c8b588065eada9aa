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


def test_decode_driver_emulated():
    from emu import emu_lib
    from lvsr_amd import synthetic, decode
    from lvsr_amd.bricks.recognizer import SpeechRecognizer
    cfg = dict(input_dim=5, num_phonemes=6, dims_bidir=[4], dim_dec=5, dim_matcher=6, attention_type="content",
               post_merge_dims=None, embed_outputs=True, data_prepend_eos=False)
    rec = SpeechRecognizer(device="cpu", params=synthetic.make_params(cfg, seed=5), lib=emu_lib(), net_config=cfg)
    rng = numpy.random.RandomState(0)
    utts = [(rng.normal(size=(9, 5)).astype(numpy.float32), [1, 2, 5]), (rng.normal(size=(7, 5)).astype(numpy.float32), [3, 5])]
    rep = decode.search(rec, utts, beam_size=3, char_discount=0.3, to_words=lambda ls: ["w%d" % l for l in ls if l != 5])
    assert len(rep["per_utterance"]) == 2 and 0.0 <= rep["cer"] and "wer" in rep
    for row in rep["per_utterance"]:
        assert numpy.isfinite(row["groundtruth_cost"])
        assert row["char_errors"] == ER.edit_distance(row["groundtruth"], row["recognized"])
