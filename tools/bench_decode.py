"""Decode benchmark (BASELINE.json configs[4], reported in DESIGN.md; not the bench.py metric): beam search width 16 with a
synthetic character FST language model (shallow fusion) over synthetic WSJ-shape utterances, WSJ-base weights, window_around_median
prior (exp/wsj/decode.sh settings: lm.weight 0.5, no_transition_cost 20, char_discount 1.0, before 10 / after 100)."""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import numpy, torch
from lvsr_amd import spec, synthetic, lm as LM
from lvsr_amd.bricks.recognizer import SpeechRecognizer
from lvsr_amd.search import CandidateNotFoundError

ap = argparse.ArgumentParser()
ap.add_argument("--utts", type=int, default=8)
ap.add_argument("--frames", type=int, default=800)
ap.add_argument("--beam", type=int, default=16)
ap.add_argument("--no-lm", action="store_true")
ap.add_argument("--host-lm", action="store_true", help="host FST walk (memoised) instead of the device kernel")
args = ap.parse_args()
cfg = spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100))
cfg["max_decoded_length_scale"] = 3.0
rec = SpeechRecognizer(device="cuda:0", params=synthetic.make_params(cfg, seed=10, scale=1.0), net_config=cfg)
if not args.no_lm:
    fst, cmap = LM.char_ngram_fst(33, seed=7)
    if args.host_lm:
        rec.set_language_model(LM.FSTLanguageModel(fst, nn_char_map=cmap, no_transition_cost=20.0, weight=0.5))
    else:
        rec.set_language_model(LM.DeviceFSTLanguageModel(fst, "cuda:0", nn_char_map=cmap, no_transition_cost=20.0, weight=0.5))
rec.init_beam_search(args.beam)
rng = numpy.random.RandomState(1234)
done, steps, t0 = 0, 0, None
for i in range(args.utts + 1):
    x = rng.normal(size=(args.frames, 40)).astype(numpy.float32)
    if i == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        outs, costs = rec.beam_search({"recordings": x}, char_discount=1.0, round_to_inf=1e9, stop_on="optimistic_future_cost")
        n = len(outs[0])
    except CandidateNotFoundError:
        n = 0
    if i >= 1:
        done += 1; steps += n
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(metric="beam-search decode", utterances=done, beam=args.beam, lm=not args.no_lm, frames_per_utt=args.frames,
                      sec_per_utt=dt / done, frames_per_sec=done * args.frames / dt, mean_best_len=steps / done)))
