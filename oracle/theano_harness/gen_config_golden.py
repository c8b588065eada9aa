#!/usr/bin/env python3
"""Generate tests/golden/configs.json.gz: what the REFERENCE's own loader (lvsr/config.py `Configuration`: parent chains,
recursive merge, dotted overrides, stage expansion) makes of every YAML under exp/*/configs and lvsr/configs, run from the
scratch copy (make_scratch.py; the loader needs two Python-2 / old-PyYAML patches listed there).  Python objects named by
the YAML tags are reduced to their class names so the result is plain JSON.  TEST INFRASTRUCTURE ONLY."""
import glob
import json
import os
import sys

import types

os.environ.setdefault("LVSR", "/root/reference")
from lvsr.config import Configuration

# the dataset classes two configs name live in lvsr.datasets, which needs Fuel / h5py / cPickle to import; only their NAMES
# matter to the loader, so name-only stand-ins are registered under the same module path
_pkg, _mod = types.ModuleType("lvsr.datasets"), types.ModuleType("lvsr.datasets.h5py")
_pkg.__path__ = []
for _n in ("H5PYAudioDataset", "H5PYAudioDatasetTimit"):
    setattr(_mod, _n, type(_n, (object,), {"__module__": "lvsr.datasets.h5py"}))
_pkg.h5py = _mod
sys.modules.setdefault("lvsr.datasets", _pkg)
sys.modules.setdefault("lvsr.datasets.h5py", _mod)

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden", "configs.json.gz")
REF = os.environ["LVSR"]


def plain(x):
    if isinstance(x, dict):
        return {str(k): plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [plain(v) for v in x]
    if isinstance(x, type):
        return {"__class__": x.__name__}
    if x is None or isinstance(x, (bool, int, float, str)):
        return x
    out = {"__instance__": type(x).__name__}
    if hasattr(x, "num_pieces"):
        out["num_pieces"] = x.num_pieces
    return out


def main():
    res = {}
    files = sorted(glob.glob(os.path.join(REF, "exp/*/configs/*.yaml")) + glob.glob(os.path.join(REF, "lvsr/configs/*.yaml")))
    changes = [("training.scale", "0.25"), ("net.dim_dec", "20")]
    for path in files:
        rel = os.path.relpath(path, REF)
        if rel.endswith("schema.yaml"):
            continue
        for tag, ch in (("plain", []), ("overrides", changes)):
            try:
                c = Configuration(path, None, ch)
                item = dict(config=plain(dict(c)), multi_stage=bool(c.multi_stage))
                if c.multi_stage:
                    item["stages"] = [[k, plain(v)] for k, v in c.ordered_stages.items()]
            except Exception as e:
                item = dict(error=type(e).__name__)
            res["%s|%s" % (rel, tag)] = item
    import gzip
    with gzip.open(OUT, "wt") as fh:
        json.dump(res, fh, sort_keys=True, separators=(",", ":"))
    print("wrote", OUT, len(res), "entries;", sum(1 for v in res.values() if "error" in v), "errors")


if __name__ == "__main__":
    main()
