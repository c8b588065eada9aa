"""Encoder kernels (same sources as the gfx950 build) on the CPU emulator vs the oracle."""
import numpy
import pytest
import torch
from numpy.testing import assert_allclose

from emu import emu_lib
from oracle import lvsr_oracle as O
from lvsr_amd import spec, synthetic
from lvsr_amd.params import ParameterStore, Workspace
from lvsr_amd.bricks import Encoder


def _cfg(Hs, sub, F=5):
    return dict(input_dim=F, num_phonemes=6, dims_bidir=Hs, subsample=sub, dim_dec=4, dim_matcher=7,
                attention_type="content", post_merge_dims=None, embed_outputs=True)


@pytest.mark.parametrize("Hs,sub,B,T,use_mask", [([3, 3], [1, 2], 3, 13, True), ([20], [3], 17, 7, True),
                                                   ([4, 5, 3], [2, 1, 2], 2, 9, False)])
def test_encoder_forward_backward(Hs, sub, B, T, use_mask):
    lib = emu_lib()
    cfg = _cfg(Hs, sub)
    params = synthetic.make_params(cfg, seed=3)
    batch = synthetic.make_batch(cfg, B, T, 4, seed=5, ragged=True)
    x = torch.tensor(batch["recordings"])
    m = torch.tensor(batch["recordings_mask"]) if use_mask else None

    orc = O.OracleRecognizer(cfg, params, dtype=torch.float64)
    enc_ref, mask_ref = orc.encode(x.double(), None if m is None else m.double())
    rng = numpy.random.RandomState(0)
    dy = torch.tensor(rng.normal(size=tuple(enc_ref.shape)), dtype=torch.float64)
    (enc_ref * dy).sum().backward()

    store = ParameterStore(cfg, "cpu", params)
    enc = Encoder(spec.Dims(cfg), store, lib, Workspace("cpu"), use_graph=False)
    out, out_mask = enc.apply(x, m)
    assert_allclose(out.numpy(), enc_ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    assert_allclose(out_mask.numpy(), mask_ref.numpy())
    enc.backward(dy.float())
    for name, g in store.g.items():
        if "/encoder/" not in name:
            continue
        ref = orc.p[name].grad.numpy()
        scale = max(1e-3, numpy.abs(ref).max())
        assert_allclose(g.numpy() / scale, ref / scale, atol=5e-5, rtol=0, err_msg=name)
