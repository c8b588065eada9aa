mkdir -p gpurun_out/r2k; O=gpurun_out/r2k
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json
d=json.load(open('$O/bench.json')); print('bench', d['ms_per_step'], d['value'], d['roofline']['us_per_recurrent_step'])"; tail -2 $O/bench.err
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.md > /dev/null; python tools/rocpd_timeline.py $DB > $O/timeline.txt; rm -rf $O/prof
cat $O/timeline.txt | head -14; head -16 $O/kernel_stats.md
