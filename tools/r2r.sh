# final refresh of the round-2 records: tests, bench lines, kernel trace, PMC passes
mkdir -p gpurun_out/r2r; O=gpurun_out/r2r
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo
timeout 300 python bench.py --ragged --no-cpu-baseline > $O/bench_ragged.json 2>/dev/null
timeout 600 python bench.py --workload wsj_decode --utterances 400 > $O/bench_decode.json 2> $O/bench_decode.err; head -c 600 $O/bench_decode.json; echo
timeout 300 python bench.py --workload wsj_deep --no-cpu-baseline > $O/bench_deep.json 2>/dev/null
timeout 300 python bench.py --workload timit_tiny --no-cpu-baseline > $O/bench_timit.json 2>/dev/null
for f in bench_ragged bench_deep bench_timit; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['ms_per_step'],3), round(d['value']))"; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.md > /dev/null; python tools/rocpd_timeline.py $DB > $O/timeline.txt; rm -rf $O/prof
head -12 $O/timeline.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 500 rocprofv3 --pmc $grp -d $O/pmc$i -o p -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $O/pmc$i.json 2> $O/pmc$i.err
done
DBS=$(find $O -name "*.db" | sort)
python tools/pmc_summary.py --json $O/r02_pmc_bench.json --tag wsj_base $DBS > $O/pmc_summary.md
grep enc_p $O/pmc_summary.md
find $O -name "*.db" -delete
