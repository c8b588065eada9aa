# Round 3: per-GPU batch sweep (what strong scaling over 2/4/8 GPUs at global batch 128 can be), the other workloads, one RCCL rank.
#   gpurun --timeout 1500 -- 'bash tools/r3e.sh'
mkdir -p gpurun_out/r3e; O=gpurun_out/r3e
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1', 'ms_per_step %.2f' % d['ms_per_step'], 'median %.2f' % d.get('ms_per_step_median', 0), 'frames/s %.0f' % d['value'], c.get('encoder_kernels'), 'us/rec.step %.2f' % d['roofline']['us_per_recurrent_step'])"; }
for b in 16 32 64 128; do
  timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/batch$b.json 2> $O/batch$b.err; line "batch=$b" < $O/batch$b.json
done
timeout 300 python bench.py --ragged --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/ragged.json 2> $O/ragged.err; line ragged < $O/ragged.json
timeout 300 python bench.py --force-dist --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/one_rccl_rank.json 2> $O/one_rccl_rank.err; line one_rccl_rank < $O/one_rccl_rank.json
timeout 400 python bench.py --workload wsj_deep --steps 5 --warmup 2 --no-cpu-baseline > $O/wsj_deep.json 2> $O/wsj_deep.err; line wsj_deep < $O/wsj_deep.json
timeout 300 python bench.py --workload timit_tiny --steps 20 --warmup 3 --no-cpu-baseline > $O/timit_tiny.json 2> $O/timit_tiny.err; line timit_tiny < $O/timit_tiny.json
timeout 300 python tools/probe_persist.py 512 8 1500 > $O/probe_enc_deep.txt 2>&1; cat $O/probe_enc_deep.txt
