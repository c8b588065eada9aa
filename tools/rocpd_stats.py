"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace): per-kernel calls / total / avg / min, plus
the median gap between consecutive dispatches.  Usage: python tools/rocpd_stats.py path/to/results.db [out.md]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name,start,end from kernels order by start"))
by = collections.defaultdict(list)
for n, s, e in rows:
    by[n].append(e - s)
tot = sum(sum(v) for v in by.values())
lines = ["| kernel | calls | total ms | avg us | min us | median us | % |", "|---|---|---|---|---|---|---|"]
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    lines.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (n[:70], len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3,
                                                              v2[0] / 1e3, v2[len(v2) // 2] / 1e3, 100.0 * sum(v) / tot))
gaps = collections.defaultdict(list)
for a, b in zip(rows[:-1], rows[1:]):
    gaps[(a[0].split("(")[0], b[0].split("(")[0])].append(b[1] - a[2])
lines.append("")
lines.append("| predecessor -> successor | n | median gap us | p10 | p90 |")
lines.append("|---|---|---|---|---|")
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 50:
        continue
    v = sorted(v)
    lines.append("| %s -> %s | %d | %.2f | %.2f | %.2f |" % (k[0][:40], k[1][:40], len(v), v[len(v) // 2] / 1e3, v[len(v) // 10] / 1e3,
                                                           v[9 * len(v) // 10] / 1e3))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
