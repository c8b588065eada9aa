# Round 3: the whole GPU suite, the default bench line (with the decode leg), rocprofv3 kernel stats + timeline of the benchmarked
# step, PMC passes (separate runs per counter group, nothing else traced).
#   gpurun --timeout 1800 -- 'bash tools/r3d.sh'
mkdir -p gpurun_out/r3d; O=gpurun_out/r3d
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json; tail -2 $O/bench_default.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-decode > $O/bench_prof.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.md > /dev/null; python tools/rocpd_timeline.py $DB > $O/timeline.txt; rm -rf $O/prof
head -24 $O/kernel_stats.md; head -16 $O/timeline.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 500 rocprofv3 --pmc $grp -d $O/pmc$i -o p -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-decode > $O/pmc$i.json 2> $O/pmc$i.err
  echo "pmc pass $i rc=$?"; tail -1 $O/pmc$i.err
done
DBS=$(find $O -name "*.db" | sort)
python tools/pmc_summary.py --json $O/r03_pmc_bench.json --tag wsj_base $DBS > $O/pmc_summary.md
head -12 $O/pmc_summary.md
find $O -name "*.db" -delete
