"""Data-parallel training step, world_size 2 over gloo on CPU (kernel sources run on the emulator): utterance
sharding r::world, ONE all-reduce of the flat gradient buffer, identical updates on every rank, and equality with
the single-process step on the unsharded batch (summation order differs -> tolerance)."""
import os
import socket
import sys

import numpy
import pytest
import torch
import torch.multiprocessing as mp
from numpy.testing import assert_allclose

CFG = dict(input_dim=5, num_phonemes=6, dims_bidir=[3, 3], subsample=[1, 2], dim_dec=4, dim_matcher=7,
           attention_type="content_and_conv", conv_n=2, conv_num_filters=3, post_merge_dims=[8],
           post_merge_activation="maxout2", embed_outputs=False, data_prepend_eos=False)
RULES = dict(gradient_threshold=5.0, rules=("momentum", "adadelta"), scale=0.5, momentum=0.0, decay_rate=0.9,
             epsilon=1e-6, max_norm=1.0)
B, T, L = 4, 13, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.dirname(here), os.path.join(os.path.dirname(here), "attention-lvcsr_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from emu import emu_lib
    from lvsr_amd import synthetic
    from lvsr_amd.bricks.recognizer import SpeechRecognizer
    from lvsr_amd.training import Trainer
    params = synthetic.make_params(CFG, seed=31)
    rec = SpeechRecognizer(device="cpu", params=params, lib=emu_lib(), net_config=CFG)
    tr = Trainer(rec, **RULES)
    assert tr.distributed and tr.world == world and tr.rank == rank
    costs = []
    for step in range(2):
        gb = synthetic.make_batch(CFG, B, T, L, seed=100 + step, ragged=True)
        shard = synthetic.shard_batch(gb, rank, world)
        cm = tr.train_step(shard, global_batch_size=B)
        costs.append(float(cm.sum()))
    numpy.savez(os.path.join(out_dir, "rank%d.npz" % rank), costs=numpy.array(costs), norm=tr.gradient_norm(),
                **{k.replace("/", "|"): v for k, v in rec.store.get_values().items()})
    torch.distributed.destroy_process_group()


def test_two_rank_step_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = numpy.load(str(tmp_path / "rank0.npz"))
    r1 = numpy.load(str(tmp_path / "rank1.npz"))
    names = [k for k in r0.files if k not in ("costs", "norm")]
    for k in names:
        assert (r0[k] == r1[k]).all(), "ranks diverged on %s" % k          # identical replicas
    assert r0["norm"] == r1["norm"]
    # single process on the whole batch
    from emu import emu_lib
    from lvsr_amd import synthetic
    from lvsr_amd.bricks.recognizer import SpeechRecognizer
    from lvsr_amd.training import Trainer
    params = synthetic.make_params(CFG, seed=31)
    rec = SpeechRecognizer(device="cpu", params=params, lib=emu_lib(), net_config=CFG)
    tr = Trainer(rec, distributed=False, **RULES)
    total = []
    for step in range(2):
        gb = synthetic.make_batch(CFG, B, T, L, seed=100 + step, ragged=True)
        total.append(float(tr.train_step(gb, global_batch_size=B).sum()))
    assert_allclose(r0["costs"] + r1["costs"], total, rtol=1e-5)
    assert_allclose(float(r0["norm"]), tr.gradient_norm(), rtol=1e-4)
    single = rec.store.get_values()
    for k in names:
        assert_allclose(r0[k], single[k.replace("|", "/")], rtol=1e-4, atol=1e-5, err_msg=k)
