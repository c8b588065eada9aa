mkdir -p gpurun_out/r2d; O=gpurun_out/r2d
timeout 300 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -2 $O/bench.err
for b in 32 64 128; do timeout 300 python bench.py --batch $b --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_b$b.json 2> $O/bench_b$b.err; python -c "
import json,sys
d=json.load(open('$O/bench_b$b.json')); print('B=$b', d['ms_per_step'], d['value'], d['config']['encoder_kernels'], d['roofline']['kernel'], d['roofline']['launch_us'])" ; done
for b in 32 64; do LVSR_PERSISTENT=1 timeout 300 python bench.py --batch $b --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_b${b}_persist.json 2> /dev/null; python -c "
import json,sys
d=json.load(open('$O/bench_b${b}_persist.json')); print('forced persistent B=$b', d['ms_per_step'], d['value'], d['config']['encoder_kernels'])" ; done
LVSR_PERSISTENT=0 timeout 300 python bench.py --batch 32 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_b32_steps.json 2> /dev/null; python -c "
import json,sys
d=json.load(open('$O/bench_b32_steps.json')); print('forced steps B=32', d['ms_per_step'], d['value'], d['config']['encoder_kernels'])"
echo "--- RCCL path with one rank (torchrun, --force-dist)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; cat $O/bench_dist1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dist1', d['ms_per_step'], d['config'].get('allreduce_ms'), d['config'].get('collective_backend'), d['config'].get('whole_step_graph_region'))"; tail -3 $O/bench_dist1.err
LVSR_DP_REGION=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-dist --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dist1 no region', d['ms_per_step'])"
echo "--- overlap experiment"
LVSR_OVERLAP=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('overlap=1 (no region)', d['ms_per_step'])"
LVSR_STEP_GRAPH=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('no region, no overlap', d['ms_per_step'])"
for w in wsj_deep timit_tiny; do timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; python -c "
import json,sys
d=json.load(open('$O/bench_$w.json')); print('$w', d['ms_per_step'], d['value'], d['config']['encoder_kernels'])"; done
