# Round 3: XCD-local clusters for any number of clusters (padded grid): the reference's batch of 10, batch-1 decoding.
mkdir -p gpurun_out/r3h; O=gpurun_out/r3h
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1', 'ms_per_step %.2f' % d['ms_per_step'], 'frames/s %.0f' % d['value'], c.get('encoder_kernels'), 'us/rec.step %.2f' % d['roofline']['us_per_recurrent_step'], 'decode', (d.get('decode') or {}).get('ms_per_utterance'))"; }
for b in 10 12 16; do
  timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/batch$b.json 2> $O/batch$b.err; line "batch=$b" < $O/batch$b.json
done
timeout 300 python bench.py --batch 10 --steps 10 --warmup 3 --no-cpu-baseline --no-decode --knob persist_flags=2 > $O/batch10_spread.json 2> $O/batch10_spread.err; line "batch=10 spread (round-2 placement for this batch)" < $O/batch10_spread.json
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/default.json 2> $O/default.err; line default < $O/default.json
timeout 300 python tools/bench_decode.py --utts 8 --streams 1 2>/dev/null | tail -1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
