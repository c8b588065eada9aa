"""FST language model walk (host) and shallow-fusion kernel vs oracle/lm_oracle.py (independent dense formulation;
parity with the reference's PyFST path is UNPINNED — PyFST is not installable here)."""
import numpy
import pytest
import torch
from numpy.testing import assert_allclose

from oracle import lm_oracle as LO
from lvsr_amd import lm as LM


def _pair(num_chars=5, seed=3):
    fst, cmap = LM.char_ngram_fst(num_chars, seed=seed)
    arcs = [(s, d, l, w) for s, lst in fst.arcs.items() for (l, d, w) in lst]
    dense = LO.DenseFST(arcs, fst.start, num_chars)
    model = LM.FSTLanguageModel(fst, nn_char_map=cmap, no_transition_cost=20.0, weight=0.5)
    return fst, dense, model


def test_att_text_reader_and_epsilon_closure():
    isyms = {"<eps>": 0, "a": 1, "b": 2}
    f = LM.ArcFST.from_att_text(["0 1 a a 0.5", "0 2 <eps> <eps> 1.0", "2 1 a a 0.25", "1 0 b b 0.1", "1 0.0"], isyms)
    st = f.expand({0: 0.0})
    assert set(st) == {0, 2} and abs(st[2] - 1.0) < 1e-12
    nxt = f.transition(st, 1)
    want = -numpy.log(numpy.exp(-0.5) + numpy.exp(-1.25))
    assert abs(nxt[1] - want) < 1e-12
    assert f.transition(st, 2) == {}


def test_state_walk_and_costs_match_dense_oracle():
    fst, dense, model = _pair()
    rng = numpy.random.RandomState(0)
    st = model.initial_states(3)
    vecs = [dense.initial() for _ in range(3)]
    for step in range(6):
        for b in range(3):
            assert_allclose(st["add"][b], dense.costs(vecs[b], model.remap_table, 20.0), rtol=1e-5, atol=1e-5)
            got = {int(s): w for s, w in zip(st["states"][b], st["weights"][b]) if s != LM.NOT_STATE}
            want = {i: v for i, v in enumerate(vecs[b]) if numpy.isfinite(v)}
            assert set(got) == set(want)
            for k in got:
                assert abs(got[k] - want[k]) < 1e-9
        outs = rng.randint(0, 5, size=3)
        st = model.transition(st, outs)
        vecs = [dense.step(v, model.remap_table[int(o)]) for v, o in zip(vecs, outs)]
    taken = model.take(st, [2, 0])
    assert (taken["states"][0] == st["states"][2]).all()


def run_fusion(device, lib):
    from lvsr_amd.native import ptr
    rng = numpy.random.RandomState(1)
    n, V = 7, 33
    am = rng.normal(0, 2, (n, V)).astype(numpy.float32)
    add = numpy.abs(rng.normal(0, 3, (n, V))).astype(numpy.float32)
    add[2, 5] = 1e12
    for flags in ((1, 0, 0), (0, 0, 0), (1, 1, 1), (0, 1, 0), (0, 0, 1)):
        a, l = torch.tensor(am, device=device), torch.tensor(add, device=device)
        out = torch.empty(n, V, device=device)
        lib.call("lvsr_shallow_fusion", lib.stream_for(out), ptr(a), V, ptr(l), n, V, 0.9, 0.5, flags[0], flags[1], flags[2],
                 -1.0, ptr(out))
        ref = -LO.shallow_fusion(am, add, 0.5, 0.9, *[bool(f) for f in flags])
        assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=1e-4 * max(1.0, 1e-7 * numpy.abs(ref).max()))


def test_shallow_fusion_kernel_emulated():
    from emu import emu_lib
    run_fusion("cpu", emu_lib())


@pytest.mark.gpu
def test_shallow_fusion_kernel_gpu(gpu_device):
    from lvsr_amd import native
    run_fusion(gpu_device, native.get())


# ---- beam search with shallow fusion: HIP generation step + host FST walk vs the oracle's dense formulation --------
CFG = dict(input_dim=5, num_phonemes=6, dims_bidir=[4, 3], subsample=[1, 2], dim_dec=5, dim_matcher=6,
           attention_type="content_and_conv", conv_n=2, conv_num_filters=3, prior=dict(type="window_around_median", before=2, after=3),
           post_merge_dims=[8], post_merge_activation="maxout2", embed_outputs=True, data_prepend_eos=False,
           max_decoded_length_scale=1)


def run_fused_beam(device, lib):
    from oracle import lvsr_oracle as O
    from lvsr_amd import synthetic
    from lvsr_amd.bricks.recognizer import SpeechRecognizer
    params = synthetic.make_params(CFG, seed=41)
    fst, cmap = LM.char_ngram_fst(6, seed=5)
    arcs = [(s, d, l, w) for s, lst in fst.arcs.items() for (l, d, w) in lst]
    dense = LO.DenseFST(arcs, fst.start, 6)
    model = LM.FSTLanguageModel(fst, nn_char_map=cmap, no_transition_cost=20.0, weight=0.5)
    rec = SpeechRecognizer(device=device, params=params, lib=lib, net_config=CFG)
    rec.set_language_model(model)
    orc = O.OracleRecognizer(CFG, params, dtype=torch.float32)
    lm = dict(dense=dense, remap=model.remap_table, no_transition_cost=20.0, weight=0.5)
    for seed, beam, kw in ((1, 4, dict(char_discount=0.2, stop_on="optimistic_future_cost")), (2, 3, dict(char_discount=1.0, round_to_inf=15.0))):
        x = numpy.random.RandomState(seed).normal(size=(14, 5)).astype(numpy.float32)
        rec.init_beam_search(beam)
        outs, costs = rec.beam_search({"recordings": x}, **kw)
        ref_outs, ref_costs = orc.beam_search(x, beam, lm=lm, **kw)
        assert outs == ref_outs
        assert_allclose(costs, ref_costs, rtol=1e-4, atol=1e-4)
    rec.set_language_model(None)
    outs2, _ = rec.beam_search({"recordings": x}, char_discount=0.2)
    assert outs2 == orc.beam_search(x, beam, char_discount=0.2)[0]


def test_beam_search_with_shallow_fusion_emulated():
    from emu import emu_lib
    run_fused_beam("cpu", emu_lib())


@pytest.mark.gpu
def test_beam_search_with_shallow_fusion_gpu(gpu_device):
    run_fused_beam(gpu_device, None)
