# Round 3: cluster launches sized by the device's static capacity (occupancy x CUs, no reserve): batch sweep, WSJ-deep, decode with 8
# searches in flight, the GPU suite.
mkdir -p gpurun_out/r3g; O=gpurun_out/r3g
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1', 'ms_per_step %.2f' % d['ms_per_step'], 'frames/s %.0f' % d['value'], c.get('encoder_kernels'), 'us/rec.step %.2f' % d['roofline']['us_per_recurrent_step'], 'decode', (d.get('decode') or {}).get('ms_per_utterance'))"; }
for b in 16 32 64; do
  timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/batch$b.json 2> $O/batch$b.err; line "batch=$b" < $O/batch$b.json
done
timeout 300 python bench.py --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-decode --knob cluster_reserve=32 > $O/batch32_r32.json 2> $O/batch32_r32.err; line "batch=32 reserve 32" < $O/batch32_r32.json
timeout 300 python bench.py --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-decode --knob cluster_reserve=32 > $O/batch64_r32.json 2> $O/batch64_r32.err; line "batch=64 reserve 32" < $O/batch64_r32.json
timeout 400 python bench.py --workload wsj_deep --steps 5 --warmup 2 --no-cpu-baseline > $O/wsj_deep.json 2> $O/wsj_deep.err; line wsj_deep < $O/wsj_deep.json
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/default.json 2> $O/default.err; line default < $O/default.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
