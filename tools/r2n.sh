# persistent decoder forward: parity on the GPU, then the step with and without it, then the kernel trace
mkdir -p gpurun_out/r2n; O=gpurun_out/r2n
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "recognizer_vs_reference or full_size" > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
LVSR_DEC_PERSISTENT=0 timeout 300 python bench.py --no-cpu-baseline > $O/bench_step.json 2> $O/bench_step.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench_pers.json 2> $O/bench_pers.err
for f in bench_step bench_pers; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['ms_per_step'],3), round(d['value']))" || tail -5 $O/$f.err; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.md > /dev/null; python tools/rocpd_timeline.py $DB > $O/timeline.txt; rm -rf $O/prof
head -14 $O/kernel_stats.md; head -12 $O/timeline.txt
