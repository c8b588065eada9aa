"""Decode benchmark (BASELINE.json configs[4]; `python bench.py --workload wsj_decode`): beam search width 16 with a synthetic
character-trigram FST language model (shallow fusion) over synthetic WSJ-shape utterances (T = 800 frames), WSJ-base weights
(random init, seed 10), window_around_median prior and the settings of exp/wsj/decode.sh:12-25 (lm.weight 0.5,
no_transition_cost 20, char_discount 1.0, before 10 / after 100, max length T/3, stop_on optimistic_future_cost).
Decoding has no exchange step: with N GPUs rank r decodes utterances r::N ("replicas only").

    python bench.py --workload wsj_decode [--utterances 1000] [--gpus N]
    python tools/bench_decode.py [--utts 8] [--frames 800] [--beam 16] [--no-lm] [--host-lm]      (stand-alone, one GPU)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (REPO, os.path.join(REPO, "attention-lvcsr_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy
import torch


# parameter scales of the reference-generated full-size decode fixtures (oracle/theano_harness/gen_golden.py WSJ_COND_DECODE on scale 2,
# language model seed 9: tests/golden/wsj_decode_full2.npz, wsj_decode_beam200.npz): contractive recurrences and sharp energies, on which the
# search explores real alternatives (hundreds of finished hypotheses of up to 117 characters) instead of ending on a bare <eol>
WSJ_COND_DECODE = {"transition.state_to": 0.15, "gatedrecurrent.state_to": 0.25, "energy_comp": 1.5, "transform_states": 0.5}


def build(device, beam=16, lm="device", conditioned=False):
    from lvsr_amd import spec, synthetic, lm as LM
    from lvsr_amd.bricks.recognizer import SpeechRecognizer
    cfg = spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100))
    cfg["max_decoded_length_scale"] = 3.0
    params = synthetic.make_params(cfg, seed=10, scale=2.0, scales=WSJ_COND_DECODE) if conditioned else synthetic.make_params(cfg, seed=10, scale=1.0)
    rec = SpeechRecognizer(device=device, params=params, net_config=cfg)
    if lm != "none":
        fst, cmap = LM.char_ngram_fst(33, seed=9 if conditioned else 7)
        if lm == "host":
            rec.set_language_model(LM.FSTLanguageModel(fst, nn_char_map=cmap, no_transition_cost=20.0, weight=0.5))
        else:
            rec.set_language_model(LM.DeviceFSTLanguageModel(fst, device, nn_char_map=cmap, no_transition_cost=20.0, weight=0.5))
    rec.init_beam_search(beam)
    return rec, cfg


def run(rec, utterances, frames, rank=0, world=1, warm=1):
    """Decode utterances rank::world of the seeded synthetic set; returns (seconds, utterances, frames, characters, steps)."""
    from lvsr_amd.search import CandidateNotFoundError
    done = chars = steps = 0
    t0 = None
    ids = list(range(rank, utterances, world))
    for j, i in enumerate([ids[0]] * warm + ids):
        x = numpy.random.RandomState(1234 + i).normal(size=(frames, 40)).astype(numpy.float32)
        if j == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        try:
            outs, costs = rec.beam_search({"recordings": x}, char_discount=1.0, round_to_inf=1e9, stop_on="optimistic_future_cost")
            n = len(outs[0])
        except CandidateNotFoundError:
            n = 0
        if j >= warm:
            done += 1
            chars += n
            steps += rec._beam_search.last_stats.get("positions", 0)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, done, done * frames, chars, steps


def run_concurrent(recs, utterances, frames, rank=0, world=1, warm=None):
    """The same set decoded with len(recs) searches in flight, one recognizer (own stream and workspaces, same parameters)
    each, driven round-robin from this thread: begin / advance / finish never block the host, the GPU overlaps the small
    kernels of different utterances.  Returns what `run` returns."""
    from lvsr_amd.search import CandidateNotFoundError
    kw = dict(char_discount=1.0, round_to_inf=1e9, stop_on="optimistic_future_cost")
    ids = list(range(rank, utterances, world))

    def wav(i):
        return numpy.random.RandomState(1234 + i).normal(size=(frames, 40)).astype(numpy.float32)

    for rec in recs:                                    # warm-up: workspaces, graph capture of the step
        for _ in range(2):
            try:
                rec.beam_search({"recordings": wav(ids[0])}, **kw)
            except CandidateNotFoundError:
                pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = list(ids)
    slots = [None] * len(recs)
    done = chars = steps = 0
    while pending or any(s is not None for s in slots):
        for k, rec in enumerate(recs):
            bs = rec._beam_search
            # the recognizer's own stream is made current for the calls: SpeechRecognizer then neither forks from nor joins
            # the caller's stream (which would chain the searches one behind the other through the default stream)
            with torch.cuda.stream(rec.stream):
                if slots[k] is None:
                    if not pending:
                        continue
                    x = wav(pending.pop(0))
                    slots[k] = bs.begin({"recordings": x[:, None, :]}, rec.eos_label, int(x.shape[0] / rec.max_decoded_length_scale),
                                        ignore_first_eol=rec.data_prepend_eos, **kw)
                run_ = slots[k]
                if bs.advance(run_, 8, wait=False):
                    try:
                        outs, _ = bs.finish(run_)
                        chars += len(outs[0])
                    except CandidateNotFoundError:
                        pass
                    steps += bs.last_stats.get("positions", 0)
                    done += 1
                    slots[k] = None
    torch.cuda.synchronize()
    return time.perf_counter() - t0, done, done * frames, chars, steps


REUSED = [0]          # positions (summed over utterances) whose second attention pass reused the first one's results


def run_batched(recs, utterances, frames, rank=0, world=1, batch=32):
    """The same set decoded `batch` utterances at a time in ONE set of launches per position (BeamSearch.search_batch: the beams
    of all of them are rows of the same kernels), with len(recs) such batches in flight (one recognizer + stream each: a batch
    runs as long as its longest search, the other batch fills the chip meanwhile).  Returns what `run` returns."""
    from lvsr_amd.search import CandidateNotFoundError
    kw = dict(char_discount=1.0, round_to_inf=1e9, stop_on="optimistic_future_cost")
    ids = list(range(rank, utterances, world))

    def wav(i):
        return numpy.random.RandomState(1234 + i).normal(size=(frames, 40)).astype(numpy.float32)

    wavs = {i: wav(i) for i in set(ids)}          # the synthetic utterances exist before the clock starts (host arrays, as a data
                                                  # pipeline hands them over; the copy to the device is inside the timed region)

    def start(rec, chunk):
        xs = [wavs[i] for i in chunk]
        limits = [int(x.shape[0] / rec.max_decoded_length_scale) for x in xs]
        return rec._beam_search.begin_batch(xs, rec.eos_label, limits, ignore_first_eol=rec.data_prepend_eos, **kw)

    for rec in recs:                                    # warm-up: workspaces, graph capture of the step
        with torch.cuda.stream(rec.stream):
            for _ in range(2):
                run_ = start(rec, (ids * batch)[:batch])
                while not rec._beam_search.advance(run_, 8, wait=True):
                    pass
                rec._beam_search.finish_batch(run_)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    chunks = [ids[i: i + batch] for i in range(0, len(ids), batch)]
    slots = [None] * len(recs)
    done = chars = steps = 0
    while chunks or any(s is not None for s in slots):
        for k, rec in enumerate(recs):
            bs = rec._beam_search
            with torch.cuda.stream(rec.stream):
                if slots[k] is None:
                    if not chunks:
                        continue
                    slots[k] = start(rec, chunks.pop(0))
                if bs.advance(slots[k], 8, wait=len(recs) == 1):
                    for r in bs.finish_batch(slots[k]):
                        if not isinstance(r, Exception):
                            chars += len(r[0][0])
                        done += 1
                    st = bs.last_stats
                    steps += sum(u["positions"] for u in st["per_utterance"]) if "per_utterance" in st else st.get("positions", 0)
                    REUSED[0] += st.get("reused", 0)
                    slots[k] = None
    torch.cuda.synchronize()
    return time.perf_counter() - t0, done, done * frames, chars, steps


def decode_bench(args, rank, world, local_rank, json_out=None):
    """The bench.py line of configs[4]."""
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = world > 1
    if dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    json_out = json_out or sys.stdout
    utterances = args.utterances or 1000
    frames, beam = 800, 16
    batch = int(getattr(args, "decode_batch", None) or 64)
    # batches in flight: 1 000 utterances measured 1.17 / 1.11 / 1.08 ms per utterance at 2 / 3 / 4 (batches of 64), 1.14 / 1.12 at
    # 2 / 3 x 128, 1.31 at 6 x 32
    streams = max(1, int(getattr(args, "streams", None) or 8))          # batches (or single searches) in flight: 8 measured best in round 6 (64 x 8: 0.916 ms per utterance, 64 x 4: 0.941)
    recs = [build(dev, beam)[0] for _ in range(streams)]
    rec = recs[0]
    if dist:
        torch.distributed.barrier()
    if batch > 1:
        sec, done, nframes, chars, steps = run_batched(recs, utterances, frames, rank, world, batch)
    elif streams == 1:
        sec, done, nframes, chars, steps = run(rec, utterances, frames, rank, world)
    else:
        sec, done, nframes, chars, steps = run_concurrent(recs, utterances, frames, rank, world)
    tot = torch.tensor([sec, done, nframes, chars, steps], dtype=torch.float64, device=dev)
    if dist:
        mx = tot[:1].clone()
        torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(tot)
        tot[0] = mx[0]
    sec, done, nframes, chars, steps = [float(v) for v in tot]
    if rank == 0:
        d = rec.d
        out = dict(metric="WSJ decode frames/sec: beam search width 16 + FST language model shallow fusion (whole node)",
                   value=nframes / sec, unit="frames/s", n_gpus=world, steps=int(done), warmup=1, ms_per_step=sec / max(done / world, 1) * 1e3,
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload="wsj_decode: %d synthetic utterances x %d frames, WSJ-base weights (random init), beam %d, "
                                        "window_around_median(before=10, after=100), char-trigram FST LM on the device (weight 0.5, "
                                        "no_transition_cost 20), char_discount 1.0, max length T/3, stop_on optimistic_future_cost"
                                        % (int(done), frames, beam),
                               utterances_per_sec=done / sec, sec_per_utterance=sec / max(done / world, 1), parallelism="replicas%d" % world,
                               utterances_per_launch_set=batch, searches_in_flight_per_gpu=streams * batch,
                               mean_best_hypothesis_length=chars / max(done, 1), positions_per_utterance=steps / max(done, 1),
                               us_per_position=(sec * 1e6 * world / steps if steps else None),
                               launches_per_position="one hipGraph replay per 8 positions (21 kernel nodes each: 2 x attention pass, "
                                                     "readout, fusion, select, feedback fork, GRU, FST walk, compaction) shared by the "
                                                     "%d utterances of a batch; no device->host synchronisation except one look at the "
                                                     "`done` words per replay" % batch,
                               encoder="persistent clusters, one pass over the padded batch"))
        print(json.dumps(out), file=json_out, flush=True)
    if dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=8)
    ap.add_argument("--frames", type=int, default=800)
    ap.add_argument("--beam", type=int, default=16)
    ap.add_argument("--conditioned", action="store_true", help="the parameter scales / language model of the reference-generated full-size decode fixtures")
    ap.add_argument("--no-lm", action="store_true")
    ap.add_argument("--host-lm", action="store_true", help="host FST walk (memoised) instead of the device kernel")
    ap.add_argument("--streams", type=int, default=1, help="searches (or batches) in flight (one recognizer + stream each)")
    ap.add_argument("--batch", type=int, default=1, help="utterances per set of launches (BeamSearch.search_batch)")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=INT", help="library knob (include/lvsr_hip.h LVSR_KNOB_*)")
    a = ap.parse_args()
    if a.knob:
        from lvsr_amd import native
        native.get().set_knobs(a.knob)
    kind = "none" if a.no_lm else ("host" if a.host_lm else "device")
    if a.batch > 1:
        recs = [build("cuda:0", a.beam, kind, a.conditioned)[0] for _ in range(a.streams)]
        sec, done, nframes, chars, steps = run_batched(recs, a.utts, a.frames, batch=a.batch)
    elif a.streams > 1:
        recs = [build("cuda:0", a.beam, kind, a.conditioned)[0] for _ in range(a.streams)]
        sec, done, nframes, chars, steps = run_concurrent(recs, a.utts, a.frames)
    else:
        rec, _ = build("cuda:0", a.beam, kind, a.conditioned)
        sec, done, nframes, chars, steps = run(rec, a.utts, a.frames)
    print(json.dumps(dict(metric="beam-search decode", utterances=done, beam=a.beam, lm=not a.no_lm, frames_per_utt=a.frames,
                          streams=a.streams, batch=a.batch, sec_per_utt=sec / done, frames_per_sec=nframes / sec, mean_best_len=chars / done,
                          positions_per_utt=steps / done, us_per_position=sec * 1e6 / max(steps, 1),
                          second_pass_reused=REUSED[0] / max(steps, 1))))
