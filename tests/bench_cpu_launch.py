"""TEST ONLY: bench.py's launcher / sharding / JSON plumbing on CPU ranks (gloo) with the kernel sources on the fiber emulator
(tests/hipemu) and a toy network.  bench.py itself has no CPU or emulator path; this wrapper hands its main() a stand-in backend.
    python tests/bench_cpu_launch.py --workload toy --gpus 2 ...        (tests/test_bench_launch.py)"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch

import bench


def toy_config():
    return dict(input_dim=5, num_phonemes=6, dims_bidir=[3, 3], subsample=[1, 2], dim_dec=4, dim_matcher=7,
                attention_type="content_and_conv", conv_n=2, conv_num_filters=3, post_merge_dims=[8],
                post_merge_activation="maxout2", embed_outputs=False, data_prepend_eos=False)


class EmulatorBackend(object):
    measured = False            # numbers are meaningless here: no roofline / cpu_baseline / decode legs, no graph priming
    collective = "gloo"
    workloads = {"toy": (toy_config, 4, 13, 5), "toy_strong128": (toy_config, 4, 13, 5)}
    strong_leg = True           # the `strong` sub-object (global batch sharded r::world + both one-GPU forms) on the toy network too:
    strong_global_batch = {"toy_strong128": 128}          # one step of each form (45 s per one-GPU step of 128 utterances on the emulator)

    def open(self, local_rank):
        from emu import emu_lib
        return torch.device("cpu"), emu_lib()


if __name__ == "__main__":
    bench.main(EmulatorBackend())
