# MFMA forms of attbwd_energy / filter_grad + persistent decoder: full GPU tests, bench, kernel trace
mkdir -p gpurun_out/r2o; O=gpurun_out/r2o
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', round(d['ms_per_step'],3), round(d['value']))" || tail -5 $O/bench.err
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.md > /dev/null; python tools/rocpd_timeline.py $DB > $O/timeline.txt; rm -rf $O/prof
head -24 $O/kernel_stats.md; head -12 $O/timeline.txt
