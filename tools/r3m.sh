# Round 3: rocprofv3 kernel stats + timeline of the other training workloads (WSJ-deep = configs[3], WSJ-base with a stacked decoder).
#   gpurun --timeout 1200 -- 'bash tools/r3m.sh'
mkdir -p gpurun_out/r3m; O=gpurun_out/r3m
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in wsj_deep wsj_stack2; do
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o bench -- python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-decode > $O/$w.json 2> $O/$w.err
  DB=$(find $O/prof_$w -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB $O/${w}_kernel_stats.md > /dev/null; python tools/rocpd_timeline.py $DB > $O/${w}_timeline.txt; rm -rf $O/prof_$w
  head -16 $O/${w}_kernel_stats.md; head -14 $O/${w}_timeline.txt
done
