# Round 3: default bench line + rocprofv3 kernel stats / timeline of the benchmarked step (no PMC passes: the kernel sources are
# the ones profiles/r03_pmc_bench.json is stamped with).
#   gpurun --timeout 1200 -- 'bash tools/r3n.sh'
mkdir -p gpurun_out/r3n; O=gpurun_out/r3n
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-decode > $O/bench_prof.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.md > /dev/null; python tools/rocpd_timeline.py $DB > $O/timeline.txt; rm -rf $O/prof
head -12 $O/kernel_stats.md | cut -c1-150; head -14 $O/timeline.txt
