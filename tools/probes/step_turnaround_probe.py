#!/usr/bin/env python3
"""The idle time between two training steps (GPU box).  The per-kernel timeline of the WSJ-base step shows ≈ 0.13 ms between the last
kernel of a step's graph and the first staging copy of the next: the host's turn-around (the drain returning, Python, four copies, the
graph launch).  Three ways to run the same steps, interleaved in one session:

  drain     the product default: the host blocks until a step's graph has drained, then prepares and launches the next;
  no-drain  the host never blocks: the next step's copies and graph are enqueued behind the running one;
  late      the host returns from a step at once, prepares the next (keys, staging copies behind the running graph) and blocks on the
            running graph only immediately before the next graph launch.

Prints ms per step and where the host's time goes in a step.
    python tools/probes/step_turnaround_probe.py [steps]
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "attention-lvcsr_amd")):
    sys.path.insert(0, p)
import torch

import bench
from lvsr_amd import spec, synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
from lvsr_amd.training import Trainer


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    factory, B, T, L = spec.WORKLOADS["wsj_base"]
    cfg = factory()
    rec = SpeechRecognizer(device=dev, params=synthetic.make_params(cfg, seed=10), net_config=cfg)
    trainer = Trainer(rec, **bench.TRAIN_CONF)
    staged = [{k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(cfg, B, T, L, seed=1234 + s).items()} for s in range(4)]
    lib = rec.lib
    for s in range(4):
        trainer.train_step(staged[s], global_batch_size=B)
    torch.cuda.synchronize()

    clock = dict(begin=0.0, drain=0.0)
    inner_begin, inner_after = lib._lvsr_region_begin, lib.after_graph
    mode = ["drain"]
    pending = [None]

    def region_begin(*a):
        if mode[0] == "late" and pending[0] is not None:
            t = time.perf_counter()
            pending[0].synchronize()
            clock["drain"] += time.perf_counter() - t
        t = time.perf_counter()
        rc = inner_begin(*a)
        clock["begin"] += time.perf_counter() - t
        if mode[0] == "late":
            pending[0] = torch.cuda.Event()
            pending[0].record(torch.cuda.current_stream(dev))
        return rc

    def after_graph(ref, n):
        if mode[0] != "drain":
            return
        t = time.perf_counter()
        inner_after(ref, n)
        clock["drain"] += time.perf_counter() - t
    lib._lvsr_region_begin, lib.after_graph = region_begin, after_graph

    print("| mode | ms per step | host: graph launch call | host: blocked on the drain | host: rest of train_step |")
    print("|---|---|---|---|---|")
    for rnd in range(3):
        if rnd == 2:
            # the graph without its last node, the device-to-host copy of the "step was skipped" word
            trainer._skip_host = None
            trainer._forget_graphs()
            mode[0] = "drain"
            for s in range(4):
                trainer.train_step(staged[s], global_batch_size=B)
            torch.cuda.synchronize()
            print("| (the step graph without the device-to-host copy of the skip word at its end) | | | | |")
        for m in ("drain", "no-drain", "late"):
            mode[0], pending[0] = m, None
            for s in range(8):
                trainer.train_step(staged[s % 4], global_batch_size=B)
            torch.cuda.synchronize()
            clock.update(begin=0.0, drain=0.0)
            host = 0.0
            t0 = time.perf_counter()
            for s in range(steps):
                t = time.perf_counter()
                trainer.train_step(staged[s % 4], global_batch_size=B)
                host += time.perf_counter() - t
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            print("| %s | %.3f | %.0f us | %.0f us | %.0f us |"
                  % (m, el / steps * 1e3, clock["begin"] / steps * 1e6, clock["drain"] / steps * 1e6,
                     (host - clock["begin"] - clock["drain"]) / steps * 1e6))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
