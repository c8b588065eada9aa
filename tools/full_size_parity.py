#!/usr/bin/env python3
"""Full-size gradient parity table (GPU box): for every full-size training fixture and both decoder paths (persistent cluster
kernels / step kernels), per parameter tensor
  * against the REFERENCE's own gradient elements kept in the fixture (`gsub:<name>`: 2048 sampled elements, small tensors whole),
  * against the float64 oracle's FULL gradient tensors (oracle/lvsr_oracle.py, pinned to the same fixtures),
the maximum difference relative to the tensor's maximum and the cosine.  Prints a markdown table (profiles/r05_full_size_parity.md).

    python tools/full_size_parity.py [case ...] [--no-oracle]
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "attention-lvcsr_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import json
import numpy
import torch

from lvsr_amd import synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer

CASES = ["wsj_base", "wsj_base_median", "wsj_base_ragged", "wsj_base_mean"]


PROP_PRIORS = {"prop_median": dict(type="window_around_median", before=10, after=100), "prop_mean": dict(type="window_around_mean", before=30, after=40)}
WSJ_COND_TRAIN = {"transition.state_to": 0.3, "gatedrecurrent.state_to": 0.5, "energy_comp": 2.0, "handler": 2.0, "transform_states": 0.3}


def load(case):
    if case in PROP_PRIORS:      # the batch of tests/test_gpu_properties.py::test_persistent_decoder_agrees_with_step_kernels: no fixture, oracle only
        from lvsr_amd import spec
        cfg = dict(spec.wsj_base(), prior=PROP_PRIORS[case])
        names = list(spec.parameter_shapes(cfg).keys())
        return dict(grad_names=numpy.array(names), files=[]), dict(cfg=cfg, param_seed=13, scale=1.0, scales=WSJ_COND_TRAIN, B=16, T=800, L=100,
                                                                     batch_seed=77, ragged=True)
    z = numpy.load(os.path.join(REPO, "tests", "golden", case + ".npz"), allow_pickle=False)
    return z, json.loads(str(z["meta"]))


def compare(got, ref, scale=None):
    got, ref = numpy.asarray(got, numpy.float64).ravel(), numpy.asarray(ref, numpy.float64).ravel()
    scale = float(numpy.abs(ref).max()) if scale is None else float(scale)
    rel = float(numpy.abs(got - ref).max() / max(scale, 1e-30))
    den = float(numpy.sqrt((got * got).sum() * (ref * ref).sum()))
    return rel, (float((got * ref).sum()) / den if den > 0 else 1.0)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    use_oracle = "--no-oracle" not in sys.argv
    dev = torch.device("cuda:0")
    print("| fixture | decoder kernels | vs | worst max-diff / tensor max (tensor) | worst cosine (tensor) | cost sum rel. diff | alignment argmax equal |")
    print("|---|---|---|---|---|---|---|")
    for case in (args or CASES):
        z, meta = load(case)
        params = synthetic.make_params(meta["cfg"], seed=meta["param_seed"], scale=meta["scale"], scales=meta.get("scales"))
        batch = synthetic.make_batch(meta["cfg"], meta["B"], meta["T"], meta["L"], seed=meta["batch_seed"], ragged=meta["ragged"])
        og = None
        if use_oracle:
            from oracle import lvsr_oracle as O
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            t0 = time.time()
            out, og = O.OracleRecognizer(meta["cfg"], params, dtype=torch.float64).cost_and_grads(batch)
            ocost = float(out["cost_matrix"].sum())
            oarg = out["weights"].detach().numpy().argmax(axis=2)
            sys.stderr.write("%s: float64 oracle %.0f s\n" % (case, time.time() - t0))
        for persistent in (True, False):
            rec = SpeechRecognizer(device=dev, params=params, net_config=meta["cfg"], use_persistent_decoder=persistent)
            cm = rec.cost_and_gradients(batch)
            torch.cuda.synchronize()
            rec.generator.check_persistent()
            got = rec.store.get_grads()
            w = rec.generator.last["weights"].cpu().numpy()
            cs = float(cm.double().sum())
            real = batch["labels_mask"] > 0
            names = [str(n) for n in z["grad_names"]]
            kern = "persistent" if persistent else "step"
            zfiles = z.files if hasattr(z, "files") else z["files"]
            if ("gsub:" + names[0]) in zfiles:
                worst, wcos = ("", 0.0), ("", 1.0)
                for n in names:
                    idx = synthetic.grad_sample_index(n, got[n].shape)
                    rel, cos = compare(got[n].ravel()[idx], z["gsub:" + n], z["gmax:" + n])
                    if rel > worst[1]:
                        worst = (n, rel)
                    if cos < wcos[1]:
                        wcos = (n, cos)
                print("| %s | %s | reference (sampled elements) | %.2e (%s) | %.7f (%s) | %.1e | %.4f |" % (
                    case, kern, worst[1], worst[0][-45:], wcos[1], wcos[0][-45:], abs(cs - float(z["cost_sum"])) / abs(float(z["cost_sum"])),
                    (w.argmax(axis=2) == z["weights_argmax"])[real].mean()))
            if og is not None:
                worst, wcos = ("", 0.0), ("", 1.0)
                for n in names:
                    rel, cos = compare(got[n], og[n])
                    if rel > worst[1]:
                        worst = (n, rel)
                    if cos < wcos[1]:
                        wcos = (n, cos)
                print("| %s | %s | float64 oracle (full tensors) | %.2e (%s) | %.7f (%s) | %.1e | %.4f |" % (
                    case, kern, worst[1], worst[0][-45:], wcos[1], wcos[0][-45:], abs(cs - ocost) / abs(ocost), (w.argmax(axis=2) == oarg)[real].mean()))
            sys.stdout.flush()
            del rec
        if og is not None and ("gsub:" + names[0]) in zfiles:       # the oracle against the reference's elements, for scale
            worst, wcos = ("", 0.0), ("", 1.0)
            for n in names:
                idx = synthetic.grad_sample_index(n, og[n].shape)
                rel, cos = compare(numpy.asarray(og[n]).ravel()[idx], z["gsub:" + n], z["gmax:" + n])
                if rel > worst[1]:
                    worst = (n, rel)
                if cos < wcos[1]:
                    wcos = (n, cos)
            print("| %s | (float64 oracle) | reference (sampled elements) | %.2e (%s) | %.7f (%s) | %.1e | |" % (
                case, worst[1], worst[0][-45:], wcos[1], wcos[0][-45:], abs(ocost - float(z["cost_sum"])) / abs(float(z["cost_sum"]))))


if __name__ == "__main__":
    main()
