# Round 3: WSJ-deep with the whole chip allowed for one cluster launch (H = 512, B = 8: one utterance per cluster = 256 work-groups),
# the B = 128 step again (workspace fix), GPU suite after the test / env-switch changes.
mkdir -p gpurun_out/r3f; O=gpurun_out/r3f
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1', 'ms_per_step %.2f' % d['ms_per_step'], 'frames/s %.0f' % d['value'], c.get('encoder_kernels'), 'us/rec.step %.2f' % d['roofline']['us_per_recurrent_step'])"; }
timeout 400 python bench.py --workload wsj_deep --steps 5 --warmup 2 --no-cpu-baseline --knob max_cluster_wgs=256 > $O/wsj_deep_256.json 2> $O/wsj_deep_256.err; line wsj_deep_whole_chip < $O/wsj_deep_256.json
LVSR_KNOB_MAX_CLUSTER_WGS=256 timeout 300 python tools/probe_persist.py 512 8 1500 2>&1 | head -6
timeout 300 python bench.py --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-decode > $O/batch128.json 2> $O/batch128.err; line "batch=128" < $O/batch128.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
