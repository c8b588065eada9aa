"""Summarise PMC passes of rocprofv3 (rocpd sqlite): per kernel name, mean counter value per dispatch.
Usage: python tools/pmc_summary.py [--json out.json --tag wsj_base] db1 [db2 ...] > summary.md

--json writes the record bench.py reads (profiles/r03_pmc_bench.json): per kernel "<name>@<tag>"
  hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE in bytes (rocprofv3 reports KiB; FETCH_SIZE on gfx950 counts 64 B per
  128-B request of a wide coalesced read, MI355X_MICROARCH.md "HBM": doubled as prescribed; WRITE_SIZE uncorrected),
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (per-SE aggregates as rocprofv3 sums them; ratio of the same aggregation),
  valu_busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES and wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES: both numerator and denominator
  are quad-cycles summed over the waves (MI355X_MICROARCH.md), so these ARE fractions — of the time a wave is resident, how much it
  issues vector-ALU instructions / waits.  (Round 3 divided by SQ_BUSY_CYCLES, which is not per wave: 1.88 was not a fraction.)"""
import collections, json, re, sqlite3, sys

argv = sys.argv[1:]
json_out, tag = None, "bench"
while argv and argv[0].startswith("--"):
    if argv[0] == "--json":
        json_out = argv[1]
    elif argv[0] == "--tag":
        tag = argv[1]
    argv = argv[2:]


def short(name):
    n = name.split("(")[0]
    n = n[5:] if n.startswith("void ") else n
    return n[:60]


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in argv:
    db = sqlite3.connect(path)
    cols = [c[1] for c in db.execute("pragma table_info('counters_collection')")]
    if "kernel_name" not in cols:
        print("columns:", cols)
        continue
    # one row per (dispatch, counter, dimension instance): sum the instances of a dispatch first
    per = collections.defaultdict(float)
    has_id = "dispatch_id" in cols
    q = "select kernel_name, counter_name, value%s from counters_collection" % (", dispatch_id" if has_id else "")
    for row in db.execute(q):
        name, cname, value = row[0], row[1], row[2]
        per[(short(name), cname, row[3] if has_id else len(per))] += value
    for (name, cname, _), v in per.items():
        a = acc[name][cname]
        a[0] += v
        a[1] += 1
names = sorted({c for d in acc.values() for c in d})
print("| kernel | dispatches | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k, d in sorted(acc.items(), key=lambda kv: -max(v[1] for v in kv[1].values())):
    n = max(v[1] for v in d.values())
    print("| %s | %d | " % (k, n) + " | ".join("%.4g" % (d[c][0] / d[c][1]) if c in d and d[c][1] else "-" for c in names) + " |")
if json_out:
    out = {}
    for k, d in acc.items():
        mean = {c: d[c][0] / d[c][1] for c in d if d[c][1]}
        if "FETCH_SIZE" not in mean or "WRITE_SIZE" not in mean:
            continue
        base = re.sub(r"<.*", "", k)
        # the persistent recurrent kernels read 4 B per lane on every 4th lane (64-B segments): NOT the 16-B-per-lane streaming
        # pattern the x2 calibration of the guide was made on -> raw FETCH_SIZE for them, x2 for everything else
        fx = 2.0          # the guide's gfx950 correction, applied to every kernel (the raw figure rides along as *_fetch_x1)
        rec = dict(fetch_kib=mean["FETCH_SIZE"], write_kib=mean["WRITE_SIZE"], fetch_factor=fx,
                   hbm_bytes_per_launch=(fx * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0,
                   hbm_bytes_per_launch_fetch_x1=(mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0,
                   dispatches=max(v[1] for v in d.values()),
                   source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --no-graph` of this "
                          "workload, mean over the dispatches of all layers; KiB -> bytes; FETCH_SIZE x2 (MI355X_MICROARCH.md gfx950 "
                          "correction), WRITE_SIZE as reported")
        if mean.get("SQ_BUSY_CYCLES"):
            if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
                rec["mfma_busy"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / mean["SQ_BUSY_CYCLES"]
        if mean.get("SQ_WAVE_CYCLES"):
            if "SQ_ACTIVE_INST_VALU" in mean:
                rec["valu_busy"] = mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_WAVE_CYCLES"]
            if "SQ_WAIT_ANY" in mean:
                rec["wait_frac"] = mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"]
        out["%s@%s" % (base, tag)] = rec
    # stamp: the record is valid for the sources of the ENCODER cluster kernels it is read for (bench.py refuses it otherwise)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_sha
    out["__stamp__"] = dict(csrc_sha256=csrc_sha())
    json.dump(out, open(json_out, "w"), indent=1, sort_keys=True)
