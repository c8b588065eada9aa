"""GPU probe: cProfile of beam-search decoding (host side) on WSJ-base weights."""
import cProfile, pstats, os, sys, io
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd"))
import numpy, torch
from lvsr_amd import spec, synthetic, lm as LM
from lvsr_amd.bricks.recognizer import SpeechRecognizer
cfg = spec.wsj_base(prior=dict(type="window_around_median", before=10, after=100))
cfg["max_decoded_length_scale"] = 3.0
rec = SpeechRecognizer(device="cuda:0", params=synthetic.make_params(cfg, seed=10), net_config=cfg)
fst, cmap = LM.char_ngram_fst(33, seed=7)
rec.set_language_model(LM.FSTLanguageModel(fst, nn_char_map=cmap, no_transition_cost=20.0, weight=0.5))
rec.init_beam_search(16)
rng = numpy.random.RandomState(1234)
xs = [rng.normal(size=(800, 40)).astype(numpy.float32) for _ in range(4)]
from lvsr_amd.search import CandidateNotFoundError
def run(x):
    try:
        return rec.beam_search({"recordings": x}, char_discount=1.0, stop_on="optimistic_future_cost")
    except CandidateNotFoundError:
        return None
run(xs[0])
pr = cProfile.Profile()
pr.enable()
for x in xs[1:]:
    run(x)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
