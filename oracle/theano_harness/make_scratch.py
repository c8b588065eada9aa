#!/usr/bin/env python3
"""Build a scratch copy of the reference (Theano + Blocks + lvsr) that runs on Python 3.10.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference); the GPU box
never sees it.  Nothing is copied INTO this repo: the scratch tree lives under /tmp and is used to
(a) pin `oracle/lvsr_oracle.py` and (b) generate the fixtures in tests/golden/ (see gen_golden.py).

Recipe = SURVEY.md Appendix C: alias shim (sitecustomize), stub modules, and a handful of one-line
patches applied to the *scratch copy* (Python-2-isms / numpy>=2 incompatibilities).
"""
import os
import re
import shutil
import sys

REF = os.environ.get("LVSR_REFERENCE", "/root/reference")
DST = os.environ.get("LVSR_ORACLE_SCRATCH", "/tmp/lvsr_oracle_scratch")

SITECUSTOMIZE = r'''
import collections, collections.abc, inspect, sys, types, numpy
for n in ('MutableMapping','MutableSet','Callable','Iterable','Mapping','Sequence','Hashable','Sized',
          'Container','Iterator','Set','MutableSequence'):
    if not hasattr(collections, n): setattr(collections, n, getattr(collections.abc, n))
if not hasattr(inspect, 'getargspec'):
    def getargspec(f):
        s = inspect.getfullargspec(f); return s.args, s.varargs, s.varkw, s.defaults
    inspect.getargspec = getargspec
for n, v in (('bool',bool),('int',int),('float',float),('complex',complex),('object',object),('str',str)):
    if not hasattr(numpy, n): setattr(numpy, n, v)
if not hasattr(numpy, 'sctype2char'): numpy.sctype2char = lambda d: numpy.dtype(d).char
for n, v in (('product','prod'),('cumproduct','cumprod'),('sometrue','any'),('alltrue','all'),('float_','float64'),
             ('complex_','complex128'),('unicode_','str_'),('round_','round'),('rank','ndim'),('NaN','nan'),
             ('Inf','inf'),('infty','inf')):
    if not hasattr(numpy, n): setattr(numpy, n, getattr(numpy, v))
if not hasattr(numpy, 'asscalar'): numpy.asscalar = lambda a: a.item()
import scipy.signal
from scipy.signal import _sigtools, _signaltools
m = types.ModuleType('scipy.signal.sigtools'); m._convolve2d = _sigtools._convolve2d
sys.modules['scipy.signal.sigtools'] = m
sys.modules['scipy.signal.signaltools'] = _signaltools
'''

STUBS = {
    "picklable_itertools/__init__.py": "from itertools import *\n",
    "picklable_itertools/extras.py": (
        "def equizip(*its):\n"
        "    its = [list(i) for i in its]\n"
        "    assert len(set(len(i) for i in its)) <= 1\n"
        "    return zip(*its)\n"),
    "toolz/__init__.py": (
        "import itertools\n"
        "def unique(seq, key=None):\n"
        "    seen = set()\n"
        "    for x in seq:\n"
        "        k = x if key is None else key(x)\n"
        "        if k not in seen:\n"
        "            seen.add(k); yield x\n"
        "def first(seq):\n    return next(iter(seq))\n"
        "def isdistinct(seq):\n    seq = list(seq); return len(seq) == len(set(seq))\n"
        "def interleave(seqs):\n"
        "    its = [iter(s) for s in seqs]\n"
        "    while its:\n"
        "        nxt = []\n"
        "        for it in its:\n"
        "            try:\n                yield next(it); nxt.append(it)\n"
        "            except StopIteration:\n                pass\n"
        "        its = nxt\n"),
    "toposort.py": (
        "def toposort_flatten(data, sort=True):\n"
        "    data = {k: set(v) for k, v in data.items()}\n"
        "    for k, v in data.items():\n        v.discard(k)\n"
        "    extra = set().union(*data.values()) - set(data.keys()) if data else set()\n"
        "    for e in extra:\n        data[e] = set()\n"
        "    out = []\n"
        "    while True:\n"
        "        ready = set(k for k, v in data.items() if not v)\n"
        "        if not ready:\n            break\n"
        "        out.extend(sorted(ready) if sort else ready)\n"
        "        data = {k: (v - ready) for k, v in data.items() if k not in ready}\n"
        "    assert not data, 'cyclic dependency'\n"
        "    return out\n"),
    # PyFST stand-in for lvsr/ops.py (the real one binds OpenFST, which is not installable here): `fst.read(path)` parses an
    # AT&T text acceptor ('src dst ilabel olabel weight' / 'final [weight]') with '<path>.isyms' ('symbol id' lines) and
    # exposes exactly the attributes ops.py touches: .start, .isyms.items(), fst[state] -> arcs with .ilabel/.nextstate/
    # .weight, state.final.  Only the CONTAINER is shimmed; the walk that gen_fst_golden.py records is the reference's code.
    "fst.py": (
        "class _Arc(object):\n"
        "    def __init__(self, il, ol, w, nxt):\n        self.ilabel, self.olabel, self.weight, self.nextstate = il, ol, w, nxt\n"
        "class _State(list):\n    final = float('inf')\n"
        "class _Fst(object):\n"
        "    def __init__(self):\n        self.states, self.start, self.isyms = {}, None, {}\n"
        "    def __getitem__(self, q):\n        return self.states.setdefault(q, _State())\n"
        "class SymbolTable(dict):\n"
        "    def __init__(self, eps='<eps>'):\n        dict.__init__(self); self[eps] = 0\n"
        "def read(path):\n"
        "    f = _Fst()\n"
        "    for line in open(path):\n"
        "        p = line.split()\n"
        "        if not p:\n            continue\n"
        "        if len(p) <= 2:\n            f[int(p[0])].final = float(p[1]) if len(p) == 2 else 0.0\n            continue\n"
        "        src, dst = int(p[0]), int(p[1])\n"
        "        if f.start is None:\n            f.start = src\n"
        "        f[src].append(_Arc(int(p[2]), int(p[3]), float(p[4]) if len(p) > 4 else 0.0, dst))\n"
        "        f[dst]\n"
        "    for line in open(path + '.isyms'):\n"
        "        s, i = line.split()\n        f.isyms[s] = int(i)\n"
        "    return f\n"),
    "pykwalify/__init__.py": "",
    "pykwalify/core.py": "class Core(object):\n    def __init__(self, **kw):\n        pass\n    def validate(self, **kw):\n        return True\n",
    "progressbar.py": "# imported by blocks.extensions (ProgressBar extension, never instantiated by the harness)\n",
    "fuel/__init__.py": "",
    "fuel/utils.py": (
        "def do_not_pickle_attributes(*names):\n"
        "    # fuel.utils: the named attributes are not pickled and are (re)created by load() on first access\n"
        "    def deco(cls):\n"
        "        def __getattr__(self, name):\n"
        "            if name in names and 'load' in dir(type(self)):\n"
        "                self.load()\n"
        "                return self.__dict__[name]\n"
        "            raise AttributeError(name)\n"
        "        cls.__getattr__ = __getattr__\n"
        "        return cls\n"
        "    return deco\n"),
}


def patch(path, subs, count_expected=None):
    with open(path) as f:
        src = f.read()
    total = 0
    for pat, rep in subs:
        src, n = re.subn(pat, rep, src)
        total += n
    if count_expected is not None and total != count_expected:
        raise RuntimeError("patch of %s applied %d times, expected %d" % (path, total, count_expected))
    with open(path, "w") as f:
        f.write(src)


def main():
    if os.path.exists(DST):
        shutil.rmtree(DST)
    pkg = os.path.join(DST, "pkg")
    shims = os.path.join(DST, "shims")
    os.makedirs(pkg)
    os.makedirs(shims)
    ign = shutil.ignore_patterns("__pycache__", "*.pyc")
    shutil.copytree(os.path.join(REF, "libs/Theano/theano"), os.path.join(pkg, "theano"), ignore=ign)
    shutil.copytree(os.path.join(REF, "libs/blocks/blocks"), os.path.join(pkg, "blocks"), ignore=ign)
    shutil.copytree(os.path.join(REF, "lvsr"), os.path.join(pkg, "lvsr"), ignore=ign)
    with open(os.path.join(pkg, "sitecustomize.py"), "w") as f:
        f.write(SITECUSTOMIZE)
    for rel, body in STUBS.items():
        p = os.path.join(shims, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(body)

    patch(os.path.join(pkg, "blocks/bricks/sequence_generators.py"),
          [(r"return \[costs\] \+ states\.values\(\) \+ glimpses\.values\(\)",
            "return [costs] + list(states.values()) + list(glimpses.values())")], 1)
    patch(os.path.join(pkg, "blocks/search.py"),
          [(r"states\.values\(\)\[0\]", "list(states.values())[0]"),
           (r"large_contexts\.values\(\)\[0\]", "list(large_contexts.values())[0]"),
           (r"output\[:mask\.sum\(\)\]", "output[:int(mask.sum())]")])
    patch(os.path.join(pkg, "theano/tensor/basic.py"),
          [(r"( +)x_ = theano\._asarray\(x, dtype=dtype\)\n( +)if numpy\.all\(x == x_\):\n( +)break",
            r"\1try:\n\1    x_ = theano._asarray(x, dtype=dtype)\n\1except OverflowError:\n\1    continue\n"
            r"\2if numpy.all(x == x_):\n\3break")], 1)
    # python-2-isms inside lvsr that the bricks-only import touches
    patch(os.path.join(pkg, "lvsr/bricks/recognizer.py"), [(r"\bxrange\b", "range")])
    patch(os.path.join(pkg, "lvsr/error_rate.py"), [(r"\bxrange\b", "range")])
    # Python 2 orders None below every number, so `max(args)` in FST.combine_weights ignores the `None` that FST.expand
    # passes for a state not yet in the set; Python 3 raises instead
    patch(os.path.join(pkg, "lvsr/ops.py"), [(r"m = max\(args\)", "m = max(a for a in args if a is not None)")], 1)
    # lvsr/config.py: tuple-parameter lambda (Python 2 only) and yaml.load() without a Loader (PyYAML >= 6 requires one;
    # the full loader is what PyYAML used by default when the reference was written)
    patch(os.path.join(pkg, "lvsr/config.py"),
          [(r"lambda \(k, v\): v\['number'\]", "lambda kv: kv[1]['number']"),
           (r"yaml\.load\(file_\)", "yaml.load(file_, Loader=yaml.Loader)"),
           (r"yaml\.load\(value\)", "yaml.load(value, Loader=yaml.Loader)")], 3)
    # dict views are not sequences in Python 3 (FSTTransition.initial_states hands them to numpy.pad)
    patch(os.path.join(pkg, "lvsr/bricks/language_models.py"),
          [(r"self\.transition\.pad\(states_dict\.keys\(\), NOT_STATE\)", "self.transition.pad(list(states_dict.keys()), NOT_STATE)"),
           (r"self\.transition\.pad\(states_dict\.values\(\), 0\)", "self.transition.pad(list(states_dict.values()), 0)")], 2)
    # AdvancedSubtensor.perform indexes with a LIST of index arrays; numpy >= 1.23 reads a list as ONE fancy index on axis 0
    # (the "non-tuple sequence for multidimensional indexing" deprecation, now removed): a tuple is what numpy 1.x understood
    # (reached by LMEmitter.cost -> SelectInEachRow, lvsr/bricks/language_models.py:141-168, i.e. `analyze` with a language model)
    patch(os.path.join(pkg, "theano/tensor/subtensor.py"),
          [(r"out\[0\] = inputs\[0\]\.__getitem__\(inputs\[1:\]\)", "out[0] = inputs[0].__getitem__(tuple(inputs[1:]))")], 1)
    print("scratch reference at", DST)
    print("run with: THEANO_FLAGS=device=cpu,floatX=float32,cxx=,optimizer_excluding=fusion,"
          "base_compiledir=/tmp/theano_cc PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=%s:%s python3 ..." % (pkg, shims))


if __name__ == "__main__":
    main()
