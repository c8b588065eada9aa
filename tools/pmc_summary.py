"""Summarise PMC passes of rocprofv3 (rocpd sqlite): per kernel name, mean counter value per dispatch.
Usage: python tools/pmc_summary.py db1 [db2 ...] > summary.md"""
import sqlite3, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cols = [c[1] for c in db.execute("pragma table_info('counters_collection')")]
    q = "select kernel_name, counter_name, value from counters_collection" if "kernel_name" in cols else None
    if q is None:
        print("columns:", cols); continue
    for name, cname, value in db.execute(q):
        a = acc[name.split("(")[0][:60]][cname]
        a[0] += value; a[1] += 1
names = sorted({c for d in acc.values() for c in d})
print("| kernel | dispatches | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k, d in sorted(acc.items(), key=lambda kv: -max(v[1] for v in kv[1].values())):
    n = max(v[1] for v in d.values())
    print("| %s | %d | " % (k, n) + " | ".join("%.4g" % (d[c][0] / d[c][1]) if c in d and d[c][1] else "-" for c in names) + " |")
