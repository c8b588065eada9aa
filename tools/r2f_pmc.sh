# PMC passes over the benchmarked step (separate runs per counter group, no tracing domains mixed in)
mkdir -p gpurun_out/r2f; O=gpurun_out/r2f
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 500 rocprofv3 --pmc $grp -d $O/pmc$i -o p -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $O/pmc$i.json 2> $O/pmc$i.err
  echo "pass $i rc=$?"; tail -2 $O/pmc$i.err
done
DBS=$(find $O -name "*.db" | sort)
echo $DBS
python tools/pmc_summary.py --json $O/r02_pmc_bench.json --tag wsj_base $DBS > $O/pmc_summary.md
head -20 $O/pmc_summary.md
find $O -name "*.db" -size +30M -delete
