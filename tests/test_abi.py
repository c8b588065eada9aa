"""C-ABI surface: every function declared in include/lvsr_hip.h is exported by the gfx950 library (no compute calls, no GPU
needed) and by the emulator build; argument blocks parsed from the header have the layout the compiler uses."""
import ctypes
import os
import subprocess
import sys
import tempfile

import pytest

from lvsr_amd import native

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_declares_the_hot_path_entry_points():
    structs, functions = native.parse_header()
    for name in ("lvsr_sgemm", "lvsr_pack_b", "lvsr_bigru_fwd", "lvsr_bigru_bwd", "lvsr_attdec_fwd", "lvsr_attdec_bwd",
                 "lvsr_attdec_filter_grad", "lvsr_softmax_nll", "lvsr_shallow_fusion", "lvsr_opt_step", "lvsr_fbank",
                 "lvsr_add_deltas_cmvn", "lvsr_gather_rows", "lvsr_scatter_add_rows", "lvsr_last_error"):
        assert name in functions, name
    for name in ("lvsr_bigru_fwd_args", "lvsr_bigru_bwd_args", "lvsr_attdec_args", "lvsr_attdec_bwd_args", "lvsr_opt_args",
                 "lvsr_fbank_cfg"):
        assert name in structs, name


def test_product_library_exports_every_declared_symbol():
    if not os.path.exists(native.DEFAULT_LIB):
        sys.path.insert(0, os.path.join(REPO, "attention-lvcsr_amd", "csrc"))
        import build as csrc_build
        csrc_build.build()
    lib = native.Lib(native.DEFAULT_LIB)          # raises NativeError on a missing symbol
    assert lib._lvsr_abi_version() >= 1
    assert not lib.is_emulator
    assert lib._lvsr_pack_size(256, 512) == 256 * 512
    assert lib._lvsr_bigru_persist_ws_bytes(16, 256) > 0 and lib._lvsr_bigru_persist_ws_bytes(16, 4096) == 0


def test_argument_block_layout_matches_the_c_compiler():
    structs, _ = native.parse_header()
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "lvsr_hip.h"\nint main(void) {\n'
    for name, cls in structs.items():
        src += '  printf("%s %%zu\\n", sizeof(%s));\n' % (name, name)
        for fname, _ in cls._fields_:
            src += '  printf("%s.%s %%zu\\n", offsetof(%s, %s));\n' % (name, fname, name, fname)
    src += "  return 0;\n}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "layout.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "layout")
        subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
    want = dict(line.split() for line in out if line.strip())
    for name, cls in structs.items():
        assert int(want[name]) == ctypes.sizeof(cls), name
        for fname, _ in cls._fields_:
            assert int(want["%s.%s" % (name, fname)]) == getattr(cls, fname).offset, (name, fname)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(native.NativeError):
        native.Lib(str(tmp_path / "liblvsr_hip.so"))
    import torch
    from lvsr_amd import spec
    from lvsr_amd.bricks.recognizer import SpeechRecognizer
    if os.path.exists(native.DEFAULT_LIB):
        with pytest.raises(native.NativeError):          # the product path has no CPU fallback
            SpeechRecognizer(device="cpu", net_config=spec.timit_tiny())
