"""Helper: build (if stale) and load the host-emulated build of the product kernel sources (TEST ONLY)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd", "csrc"))
import build as _build  # noqa: E402

_lib = None


def emu_lib():
    global _lib
    if _lib is None:
        from lvsr_amd import native
        path = _build.build_emu()
        _lib = native.Lib(path)
    return _lib
