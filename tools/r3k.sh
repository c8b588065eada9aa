# Round 3: unit-blocked forward kernel at H = 512 (WSJ-deep), full GPU suite, default bench
mkdir -p gpurun_out/r3k; O=gpurun_out/r3k
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1', 'ms_per_step %.2f' % d['ms_per_step'], 'frames/s %.0f' % d['value'], 'us/rec.step %.2f' % d['roofline']['us_per_recurrent_step'])"; }
timeout 400 python bench.py --workload wsj_deep --steps 5 --warmup 2 --no-cpu-baseline > $O/wsj_deep.json 2> $O/wsj_deep.err; line wsj_deep < $O/wsj_deep.json
timeout 400 python bench.py --workload wsj_deep --steps 5 --warmup 2 --no-cpu-baseline --knob persist_flags=2048 > $O/wsj_deep_noub.json 2> $O/wsj_deep_noub.err; line "wsj_deep, one unit per lane group" < $O/wsj_deep_noub.json
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/default.json 2> $O/default.err; line default < $O/default.json
timeout 300 python tools/probe_persist.py 512 8 1500 2>&1 | grep -E "persist   |one unit per|no dots   "
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
