"""Per-step timeline from a rocprofv3 rocpd database: wall span of the last training step, busy time, and the largest idle
gaps with their neighbouring kernels.  Usage: python tools/rocpd_timeline.py results.db [--list]   (--list: every kernel of the step in order)"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end from kernels order by start"))
# a step ends with opt_apply_kernel
ends = [i for i, r in enumerate(rows) if r[0].startswith("opt_apply_kernel")]
if len(ends) < 2:
    print("not enough steps"); sys.exit(0)
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
span = step[-1][2] - step[0][1]
busy = sum(e - s for _, s, e in step)
print("last step: %d kernels, wall %.2f ms, sum of kernel durations %.2f ms" % (len(step), span / 1e6, busy / 1e6))
groups = collections.OrderedDict()
def grp(n):
    n = n[5:] if n.startswith("void ") else n
    for k in ("enc_pfwd", "enc_pbwd", "enc_gates", "enc_cand", "enc_bwd_a", "enc_bwd_b", "lvsr_sgemm", "lvsr_colsum", "attdec", "attbwd", "opt_", "lvsr_pack"):
        if n.startswith(k):
            return k
    return "other"
for n, s, e in step:
    g = groups.setdefault(grp(n), [0, 0])
    g[0] += 1; g[1] += e - s
for k, (c, t) in groups.items():
    print("  %-12s %6d launches %8.2f ms" % (k, c, t / 1e6))
gaps = []
for (n0, s0, e0), (n1, s1, e1) in zip(step[:-1], step[1:]):
    gaps.append((s1 - e0, n0.split("(")[0], n1.split("(")[0]))
tot = sum(g for g, _, _ in gaps if g > 0)
print("idle between kernels: %.2f ms; top gaps:" % (tot / 1e6))
for g, n0, n1 in sorted(gaps, reverse=True)[:15]:
    print("  %8.1f us  %s -> %s" % (g / 1e3, n0[:40], n1[:40]))

if "--list" in sys.argv:
    print("every kernel of the last step (start offset us, duration us, name):")
    t0 = step[0][1]
    for n, s_, e in step:
        print("  %9.1f %8.1f  %s" % ((s_ - t0) / 1e3, (e - s_) / 1e3, (n[5:] if n.startswith("void ") else n).split("(")[0][:70]))
