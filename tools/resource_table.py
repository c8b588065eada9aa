#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy of every kernel in the given csrc files, from hipcc's -Rpass-analysis=kernel-resource-usage
(the cross-compiler: runs without a GPU).   python tools/resource_table.py [file.hip ...] [--filter substring] > profiles/rNN_resource_usage.md"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "attention-lvcsr_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*$", "", o) for o in out]


def table(path):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(REPO, "include"),
           "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark: +(?:Function )?Name: (\S+)", line)
        if m:
            cur = dict(name=m.group(1))
            rows.append(cur)
            continue
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/block\]| \[waves/SIMD\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["mangled"], r["name"] = r["name"], d
    return rows


def scratch_instructions(path):
    """kernel (mangled) -> (scratch loads+stores in the whole kernel, those inside a loop): a second compile to assembly; a basic
    block the compiler annotates with `in Loop:` lies inside a loop (for the persistent kernels: the label / time loop)."""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(REPO, "include"),
           "-S", "--cuda-device-only", "-c", path, "-o", "/dev/stdout"]
    asm = subprocess.run(cmd, capture_output=True, text=True).stdout
    out, cur, in_loop = {}, None, False
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+|\w+):\s*; @", line)
        if m:
            cur = m.group(1)
            out[cur] = [0, 0]
            in_loop = False
            continue
        if cur is None:
            continue
        if re.match(r"^\.LBB\d+_\d+:", line) or line.startswith("; %bb."):
            in_loop = "in Loop:" in line or "Loop Header:" in line or "Parent Loop" in line
        elif re.match(r"\s+scratch_(load|store)", line):
            out[cur][0] += 1
            out[cur][1] += int(in_loop)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = None
    if "--filter" in sys.argv:
        flt = sys.argv[sys.argv.index("--filter") + 1]
        args = [a for a in args if a != flt]
    files = args or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    print("| file | kernel | VGPR | AGPR | scratch B/lane | scratch instructions (in a loop) | LDS B | occupancy (waves/SIMD) |")
    print("|---|---|---|---|---|---|---|---|")
    for f in files:
        path = f if os.path.exists(f) else os.path.join(CSRC, f)
        rows = table(path)
        si = scratch_instructions(path) if any(r.get("ScratchSize") for r in rows) else {}
        for r in rows:
            if flt and flt not in r["name"]:
                continue
            n = si.get(r["mangled"], [0, 0])
            print("| %s | `%s` | %s | %s | %s | %d (%d) | %s | %s |" % (os.path.basename(path), r["name"], r.get("VGPRs", "?"), r.get("AGPRs", "?"),
                                                                  r.get("ScratchSize", "?"), n[0], n[1], r.get("LDS Size", "?"), r.get("Occupancy", "?")))


if __name__ == "__main__":
    main()
