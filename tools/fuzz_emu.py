"""CPU fuzz: random small network configurations / batch shapes through the EMULATED kernel sources against the float64
oracle (cost matrix, alignments, every gradient).  Not part of the test suite (minutes); `python tools/fuzz_emu.py [n] [seed]`."""
import os, sys, time, traceback
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "attention-lvcsr_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy, torch
from emu import emu_lib
from oracle import lvsr_oracle as O
from lvsr_amd import synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
from test_emu_recognizer import check_against


def random_case(rng):
    n_layers = rng.randint(1, 4)
    sub = [int(rng.choice([1, 1, 2, 3])) for _ in range(n_layers)]
    conv = rng.rand() < 0.7
    prior = None
    if conv:
        kind = rng.choice(["none", "expanding", "window_around_median", "window_around_mean"])
        if kind == "expanding":
            prior = dict(type="expanding", initial_begin=0, initial_end=int(rng.randint(2, 6)), min_speed=float(rng.choice([0.0, 0.5])),
                         max_speed=float(rng.choice([1.0, 2.5, 1e4])))
        elif kind != "none":
            prior = dict(type=str(kind), before=int(rng.randint(1, 4)), after=int(rng.randint(1, 5)))
    cfg = dict(input_dim=int(rng.randint(1, 9)), num_phonemes=int(rng.randint(3, 40)),
               dims_bidir=[int(rng.randint(1, 40)) for _ in range(n_layers)], subsample=sub,
               dim_dec=int(rng.randint(1, 36)), dim_matcher=int(rng.randint(1, 70)),
               attention_type="content_and_conv" if conv else "content", embed_outputs=bool(rng.rand() < 0.5),
               data_prepend_eos=False)
    if conv:
        cfg.update(conv_n=int(rng.randint(1, 6)), conv_num_filters=int(rng.choice([1, 2, 3, 4, 5, 8, 10, 11, 16])), prior=prior,
                   energy_normalizer=str(rng.choice(["softmax", "softmax", "logistic", "relu"])))
    if rng.rand() < 0.6:
        cfg.update(post_merge_dims=[2 * int(rng.randint(1, 12))], post_merge_activation=str(rng.choice(["maxout2", "rectifier", "tanh"])))
    if cfg["embed_outputs"] and rng.rand() < 0.4:
        cfg["dim_output_embedding"] = int(rng.randint(1, 20))
    if rng.rand() < 0.25:
        cfg.update(bottom_dims=[int(rng.randint(1, 12))], bottom_activation=str(rng.choice(["rectifier", "tanh"])))
    B, L = int(rng.randint(1, 19)), int(rng.randint(1, 8))
    T = int(numpy.prod(sub)) * int(rng.randint(1, 7)) + int(rng.randint(0, 3))
    return cfg, B, T, L


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None            # re-run one case ...
    dtype = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.float64   # ... optionally against the f32 oracle
    rng = numpy.random.RandomState(seed)
    bad = 0
    for k in range(n):
        cfg, B, T, L = random_case(rng)
        scale, ragged = float(rng.choice([0.5, 1.0, 2.0])), bool(rng.rand() < 0.7)
        if only is not None and k != only:
            continue
        t0 = time.time()
        try:
            params = synthetic.make_params(cfg, seed=100 + k, scale=scale)
            batch = synthetic.make_batch(cfg, B, T, L, seed=200 + k, ragged=ragged)
            orc = O.OracleRecognizer(cfg, params, dtype=dtype)
            out, grads = orc.cost_and_grads(batch)
            if not numpy.isfinite(out["cost_matrix"].detach().numpy()).all():
                print("[%d] skipped: the oracle itself is not finite (%s)" % (k, cfg.get("energy_normalizer")))
                continue
            rec = SpeechRecognizer(device="cpu", params=params, lib=emu_lib(), net_config=cfg)
            cm = rec.cost_and_gradients(batch)
            try:
                check_against(rec, cm, None, out, grads, tol=2.0)
            except AssertionError:
                # conditioning check: if the oracle itself moves this much between float32 and float64 arithmetic, the case
                # amplifies rounding (large random weights, alignment near a tie) and says nothing about the kernels
                o32, _ = O.OracleRecognizer(cfg, params, dtype=torch.float32).cost_and_grads(batch)
                ref_w = out["weights"].detach().numpy()
                drift = float(numpy.abs(o32["weights"].detach().numpy() - ref_w).max())
                mine = float(numpy.abs(rec.generator.last["weights"].cpu().numpy() - ref_w).max())
                if mine <= 10.0 * drift:
                    print("[%d] rounding-amplified: kernel vs f64 oracle %.1e, f32 vs f64 oracle %.1e (errors grow step by step "
                          "from 1e-6), skipped" % (k, mine, drift), flush=True)
                    continue
                raise
            print("[%d] ok  B=%d T=%d L=%d %.1fs  %s" % (k, B, T, L, time.time() - t0, {k2: cfg[k2] for k2 in ("dims_bidir", "subsample", "dim_dec", "dim_matcher", "attention_type")}), flush=True)
        except Exception as e:
            bad += 1
            print("[%d] FAIL B=%d T=%d L=%d cfg=%r\n%s" % (k, B, T, L, cfg, "".join(traceback.format_exception_only(type(e), e))[:1500]), flush=True)
    print("failures:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
