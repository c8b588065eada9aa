mkdir -p gpurun_out/r2e; O=gpurun_out/r2e
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; python -c "
import json,sys
d=json.load(open('$O/bench_dist1.json')); print('dist1 region', d['ms_per_step'], d['config'].get('allreduce_ms'), d['config'].get('collective_backend'), d['config'].get('whole_step_graph_region'))"; grep -i "error\|abort" $O/bench_dist1.err | head -5
LVSR_DP_REGION=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-dist --steps 20 --warmup 3 --no-cpu-baseline 2>$O/bench_dist1_noregion.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dist1 no region', d['ms_per_step'])"
timeout 300 python bench.py --scaling strong --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('strong N=1', d['ms_per_step'], d['value'], d['config']['global_batch'], d['config']['encoder_kernels'])"
