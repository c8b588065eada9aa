mkdir -p gpurun_out/r2s; O=gpurun_out/r2s
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.md > /dev/null; python tools/rocpd_timeline.py $DB > $O/timeline.txt; rm -rf $O/prof
head -30 $O/kernel_stats.md; head -14 $O/timeline.txt
