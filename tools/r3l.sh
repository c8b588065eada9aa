# Round 3, final measurement pass: the whole GPU suite, the default bench line (with the decode leg and the CPU baseline), rocprofv3
# kernel stats + timeline of the benchmarked step, PMC passes (separate runs per counter group, nothing else traced), then the
# other workloads / per-GPU batches, the GEMM shapes of the step and the phase clocks of the persistent decoder kernels.
#   gpurun --timeout 2400 -- 'bash tools/r3l.sh'
mkdir -p gpurun_out/r3l; O=gpurun_out/r3l
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json; tail -2 $O/bench_default.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-decode > $O/bench_prof.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.md > /dev/null; python tools/rocpd_timeline.py $DB > $O/timeline.txt; rm -rf $O/prof
head -14 $O/kernel_stats.md; head -12 $O/timeline.txt
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
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1', 'ms_per_step %.2f' % d['ms_per_step'], 'median %.2f' % d.get('ms_per_step_median', 0), 'frames/s %.0f' % d['value'], c.get('encoder_kernels'), 'us/rec.step %.2f' % d['roofline']['us_per_recurrent_step'])"; }
for b in 10 32 64 128; do
  timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/batch$b.json 2> $O/batch$b.err; line "batch=$b" < $O/batch$b.json
done
timeout 300 python bench.py --ragged --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/ragged.json 2> $O/ragged.err; line ragged < $O/ragged.json
timeout 300 python bench.py --force-dist --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/one_rccl_rank.json 2> $O/one_rccl_rank.err; line one_rccl_rank < $O/one_rccl_rank.json
timeout 400 python bench.py --workload wsj_deep --steps 5 --warmup 2 --no-cpu-baseline > $O/wsj_deep.json 2> $O/wsj_deep.err; line wsj_deep < $O/wsj_deep.json
timeout 400 python bench.py --workload wsj_stack2 --steps 6 --warmup 2 --no-cpu-baseline > $O/wsj_stack2.json 2> $O/wsj_stack2.err; line wsj_stack2 < $O/wsj_stack2.json
timeout 300 python bench.py --workload timit_tiny --steps 20 --warmup 3 --no-cpu-baseline > $O/timit_tiny.json 2> $O/timit_tiny.err; line timit_tiny < $O/timit_tiny.json
timeout 300 python tools/gemm_shapes.py > $O/gemm_shapes.txt 2>&1; tail -1 $O/gemm_shapes.txt
timeout 300 python tools/probes/gemm_k_sweep.py > $O/gemm_k_sweep.txt 2>&1; tail -3 $O/gemm_k_sweep.txt
timeout 300 python tools/probe_decoder_persist.py wsj_base > $O/decoder_fwd_phase_clock.txt 2>&1; timeout 300 python tools/probe_decoder_persist_bwd.py wsj_base > $O/decoder_bwd_phase_clock.txt 2>&1; tail -16 $O/decoder_bwd_phase_clock.txt
