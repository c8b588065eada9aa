# Round 3: forward loader waves — bench, all workloads' sanity, full GPU suite
mkdir -p gpurun_out/r3j; O=gpurun_out/r3j
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1', 'ms_per_step %.2f' % d['ms_per_step'], 'frames/s %.0f' % d['value'], 'us/rec.step %.2f' % d['roofline']['us_per_recurrent_step'], 'decode', (d.get('decode') or {}).get('ms_per_utterance'))"; }
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/default.json 2> $O/default.err; line default < $O/default.json
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --knob persist_flags=256 > $O/nostage.json 2> $O/nostage.err; line "owners fetch (round-2 form)" < $O/nostage.json
for b in 32 64; do timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/batch$b.json 2> $O/batch$b.err; line "batch=$b" < $O/batch$b.json; done
timeout 400 python bench.py --workload wsj_deep --steps 5 --warmup 2 --no-cpu-baseline > $O/wsj_deep.json 2> $O/wsj_deep.err; line wsj_deep < $O/wsj_deep.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
