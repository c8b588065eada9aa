#!/usr/bin/env python3
"""Build the C-ABI library of the hot path.

  python build.py            -> hipcc --offload-arch=gfx950  => ../lvsr_amd/liblvsr_hip.so   (product)
  python build.py --emu      -> host clang++ + tests/hipemu  => tests/hipemu/liblvsr_emu.so  (TEST ONLY:
                                same sources on CPU fibers, see tests/hipemu/hip/hip_runtime.h)
  python build.py --probes   -> hipcc -DLVSR_PROBES          => tools/probes/liblvsr_hip_probes.so  (MEASUREMENT ONLY: the
                                timing ablations of csrc/persist.h that produce wrong results; tools/probe_persist.py)
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SRCS = sorted(glob.glob(os.path.join(HERE, "*.hip")))
HDRS = sorted(glob.glob(os.path.join(HERE, "*.h"))) + sorted(glob.glob(os.path.join(REPO, "include", "*.h")))
OUT = os.path.join(os.path.dirname(HERE), "lvsr_amd", "liblvsr_hip.so")
PROBES_OUT = os.path.join(REPO, "tools", "probes", "liblvsr_hip_probes.so")
EMU_OUT = os.path.join(REPO, "tests", "hipemu", "liblvsr_emu.so")
EMU_INC = os.path.join(REPO, "tests", "hipemu")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, probes=False):
    deps = SRCS + HDRS + [os.path.abspath(__file__)]
    OUT = PROBES_OUT if probes else globals()["OUT"]
    if not force and not _stale(OUT, deps):
        return OUT
    objs = []
    bdir = os.path.join(HERE, "build_probes" if probes else "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for s in SRCS:
        o = os.path.join(bdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + HDRS):
            cmd = [os.path.join(ROCM, "bin", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                   "-I", os.path.join(REPO, "include"), "-c", s, "-o", o]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            if probes:
                cmd.insert(1, "-DLVSR_PROBES")
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    cmd = [os.path.join(ROCM, "bin", "hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    subprocess.check_call(cmd)
    return OUT


def build_emu(force=False):
    deps = SRCS + HDRS + [os.path.join(EMU_INC, "hip", "hip_runtime.h")] + sorted(glob.glob(os.path.join(EMU_INC, "*.cpp")))
    if not force and not _stale(EMU_OUT, deps):
        return EMU_OUT
    cxx = os.path.join(ROCM, "lib", "llvm", "bin", "clang++")
    bdir = os.path.join(EMU_INC, "build")
    os.makedirs(bdir, exist_ok=True)
    objs, procs = [], []
    for s in SRCS + sorted(glob.glob(os.path.join(EMU_INC, "*.cpp"))):
        o = os.path.join(bdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + HDRS + [os.path.join(EMU_INC, "hip", "hip_runtime.h")]):
            cmd = [cxx, "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-Wno-unused-value", "-Wno-psabi", "-I", EMU_INC,
                   "-I", os.path.join(REPO, "include"), "-c", s, "-o", o]
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("emu compile failed on %s" % s)
    subprocess.check_call([cxx, "-shared", "-fPIC", "-o", EMU_OUT] + objs)
    return EMU_OUT


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--emu" in sys.argv:
        print(build_emu(force))
    else:
        print(build(force, verbose="--verbose" in sys.argv, probes="--probes" in sys.argv))
