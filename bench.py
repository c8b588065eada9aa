#!/usr/bin/env python3
"""Benchmark of the hot path: encoder+attention+decoder TRAINING frames/sec on WSJ-shape synthetic fbank batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload wsj_base|wsj_deep|timit_tiny]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward + backward of the whole recognizer on one minibatch already resident in HBM, the (RCCL) sum
all-reduce of the flat gradient buffer when N > 1, and the fused optimiser step (clip -> scale -> AdaDelta ->
max-norm -> remove-not-finite).  Weak scaling: every rank processes B utterances per step (global batch N*B,
rank r takes utterances r::N of the seeded global batch).  value = real (unpadded) input frames of all ranks per
second.  Rank 0 prints ONE JSON line on stdout; everything else goes to stderr.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "attention-lvcsr_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy
import torch


def log(*a):
    print(*a, file=sys.stderr, flush=True)


TRAIN_CONF = dict(gradient_threshold=100.0, rules=("momentum", "adadelta"), scale=0.1, momentum=0.0, decay_rate=0.95,
                  epsilon=1e-8, max_norm=1.0)        # exp/wsj/configs/wsj_jan_new.yaml training/regularization sections


def recurrent_kernel_probe(rec, dims, T, B):
    """Average launch duration of the dominant kernel (enc_bwd_b_kernel: dh_prev = ... + dpre_r @ Whg[:, H:]^T, both
    directions of one layer per launch) measured with HIP events on the recognizer's own stream over T back-to-back
    graph-replayed launches (kernel_mask = 2; the figure therefore includes the dependent-launch boundary, which IS the
    cost of this latency-bound kernel)."""
    lib, ws, enc = rec.lib, rec.ws, rec.encoder
    H = dims.Hs[0]
    pk = enc._packed(0)
    p = rec.store.p
    nf, nb = enc._names(0, "forward"), enc._names(0, "backward")
    Bp = (B + 15) // 16 * 16
    bufs = dict(y=ws.get("enc0.y", (T, B, 2 * H)), u=ws.get("enc0.u", (T, B, 2 * H)), r=ws.get("enc0.r", (T, B, 2 * H)),
                c=ws.get("enc0.c", (T, B, 2 * H)), dy=ws.get("probe.dy", (T, B, 2 * H)), dxg=ws.get("enc0.dxg", (T, B, 6 * H)),
                dh_ws=ws.get("enc0.dh", (12 * Bp * H,)))
    fields = dict(mask=None, WhhT_p=[pk["WhhT"][0], pk["WhhT"][1]], WhgT_p=[pk["WhgT"][0], pk["WhgT"][1]],
                  h0=[p[nf["h0"]], p[nb["h0"]]], dh0=[ws.get("probe.dh0a", (H,)), ws.get("probe.dh0b", (H,))],
                  sub=1, T=T, B=B, H=H, kernel_mask=2, **bufs)
    times = []
    with torch.cuda.stream(rec.stream):
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(rec.stream)
            lib.run("lvsr_bigru_bwd", "lvsr_bigru_bwd_args", bufs["dxg"], True, **fields)
            e1.record(rec.stream)
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e-3 / T)
    avg = sorted(times[1:])[len(times[1:]) // 2]
    flops = 2.0 * B * H * H * 2                  # (B,H) x (H,H) per direction, both directions in one launch
    return avg, flops


def gemm_probe(rec, dims, T, B):
    """The step's largest dense contraction (a layer's gate projection, (T*B, 2H_prev) x (2H_prev, 2H)) timed alone with HIP
    events on the recognizer's stream: the MFMA-bound part of the path, reported beside the latency-bound dominant kernel."""
    lib = rec.lib
    M, K, N = T * B, 2 * dims.Hs[0], 2 * dims.Hs[0]
    dev = rec.device
    A, Bm, C = (torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev))
    with torch.cuda.stream(rec.stream):
        for _ in range(3):
            lib.sgemm(A, Bm, C)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(rec.stream)
        for _ in range(20):
            lib.sgemm(A, Bm, C)
        e1.record(rec.stream)
        e1.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / 20
    return dict(kernel="lvsr_sgemm128_kernel", shape=[M, N, K], launch_us=sec * 1e6, achieved=2.0 * M * N * K / sec / 1e12, unit="TFLOP/s")


def cpu_baseline(cfg, params, B, T, L):
    """The CPU oracle (torch fp32 restatement of the reference's algorithm, oracle/lvsr_oracle.py) timed on this
    box's host cores on whole minibatches of the same workload (forward + backward)."""
    from oracle import lvsr_oracle as O
    from lvsr_amd import synthetic
    batch = synthetic.make_batch(cfg, B, T, L, seed=1234)
    orc = O.OracleRecognizer(cfg, params, dtype=torch.float32)
    ncores = min(8, os.cpu_count() or 1)        # the per-step matrices are small: more threads only add overhead
    torch.set_num_threads(ncores)
    # bounded sample: whole minibatches until ~12 s of CPU work have been done (at least one, at most four)
    reps, t0 = 0, time.time()
    while reps < 1 or (time.time() - t0 < 12.0 and reps < 4):
        orc.cost_and_grads(batch)
        reps += 1
    dt = time.time() - t0
    return dict(value=reps * B * T / dt, unit="frames/s", cores=ncores, kind="port",
                sample="%d forward+backward passes over one %dx%d-frame minibatch (%s) = %.1f s of CPU work; torch-CPU fp32 "
                       "restatement of the reference's Theano graph" % (reps, B, T, "same synthetic batch shape", dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="wsj_base")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--ragged", action="store_true",
                    help="secondary run of SURVEY.md 8(d): utterance lengths ~U{T/2..T}, zero padded; counts real frames only")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..."
                         % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = world > 1
    if dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from lvsr_amd import spec, synthetic
    from lvsr_amd.bricks.recognizer import SpeechRecognizer
    from lvsr_amd.training import Trainer

    factory, B, T, L = spec.WORKLOADS[args.workload]
    cfg = factory()
    dims = spec.Dims(cfg)
    params = synthetic.make_params(cfg, seed=10)
    rec = SpeechRecognizer(device=dev, params=params, net_config=cfg, use_graph=not args.no_graph)
    trainer = Trainer(rec, distributed=dist, **TRAIN_CONF)
    nsteps = args.steps + args.warmup
    # synthetic global batches, seeded identically on every rank; rank r keeps utterances r::world; staged in HBM
    nstage = min(nsteps, 4)
    staged = []
    for s in range(nstage):
        gb = synthetic.make_batch(cfg, B * world, T, L, seed=1234 + s, ragged=args.ragged)
        sh = synthetic.shard_batch(gb, rank, world)
        staged.append({k: torch.from_numpy(v).to(dev) for k, v in sh.items()})
    # real (unpadded) frames per step: B*T per rank with the default all-ones masks; the mean over the staged batches when ragged
    frames_per_step = float(sum(float(b["recordings_mask"].sum()) for b in staged)) / len(staged) * world
    torch.cuda.synchronize()

    def barrier():
        if dist:
            torch.distributed.barrier()

    # setup, not warm-up: the first step of a shape allocates the workspaces, the second captures the whole-step hipGraph
    # (lvsr_amd.native.Region); whatever --warmup says, the timed steps are replays.  Reported as config.priming_steps.
    PRIME = 2 if not args.no_graph else 0
    for s in range(PRIME):
        trainer.train_step(staged[s % nstage], global_batch_size=B * world)
    costs = []
    for s in range(args.warmup):
        cm = trainer.train_step(staged[s % nstage], global_batch_size=B * world)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        cm = trainer.train_step(staged[(args.warmup + s) % nstage], global_batch_size=B * world)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t[0])
    last_cost = float(cm.sum())
    assert numpy.isfinite(last_cost), "training diverged in the benchmark"
    ms = elapsed / args.steps * 1e3
    value = frames_per_step * args.steps / elapsed

    if rank == 0:
        train_flop_per_frame = {"wsj_base": 22.730e6, "wsj_deep": 143.43e6, "timit_tiny": 1.382e6}[args.workload]
        avg_launch, flops = recurrent_kernel_probe(rec, dims, T, B)
        peak = 157.3                                                   # TFLOP/s fp32 MFMA (MI355X_MICROARCH.md)
        # HBM-side bytes per launch of this kernel from rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, KiB, uncorrected;
        # profiles/r01_pmc_small_eager_step.md, H=256 B=16): 599 + 189 KiB vs 0.56 MB algorithmic (weights 512 KiB + rows)
        traffic = (599.0 + 189.1) * 1024 if (dims.Hs[0] == 256 and B == 16) else None
        roof = dict(bound="mfma", kernel="enc_bwd_b_kernel", achieved=flops / avg_launch / 1e12, peak=peak, unit="TFLOP/s",
                    frac=flops / avg_launch / 1e12 / peak, traffic=traffic, launch_us=avg_launch * 1e6, flops_per_launch=flops,
                    whole_step_tflops=value / world * train_flop_per_frame / 1e12,
                    whole_step_frac=value / world * train_flop_per_frame / 1e12 / peak)
        roof["dense_gemm"] = gemm_probe(rec, dims, T, B)
        roof["dense_gemm"]["frac"] = roof["dense_gemm"]["achieved"] / peak
        out = dict(metric="encoder+attention+decoder training frames/sec (whole node)", value=value, unit="frames/s",
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms, higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload="%s: B=%d utterances x T=%d frames x F=%d fbank per GPU, L=%d labels; %s" % (
                       args.workload, B, T, dims.F, L, "x".join(str(h) for h in dims.Hs) + " BiGRU subsample " +
                       str(dims.subsample) + ", " + cfg["attention_type"] + " attention, %d-unit GRU decoder" % dims.D),
                       global_batch=B * world, per_gpu_batch=B, frames_per_step=frames_per_step,
                       ragged=bool(args.ragged), parallelism="dp%d" % world, optimizer="clip100+adadelta+maxnorm1", hip_graph=not args.no_graph,
                       priming_steps=PRIME,
                       final_cost_per_utterance=last_cost / B),
                   roofline=roof)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, params, B, T, L)
        print(json.dumps(out), flush=True)
    barrier()
    if dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
