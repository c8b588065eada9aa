# First GPU call of round 3: the spill-free persistent decoder backward and the one-exchange-per-step encoder kernels against the defaults.
#   gpurun --timeout 1500 -- 'bash tools/r3a.sh'   (≈15-20 min of box time)
mkdir -p gpurun_out/r3a; O=gpurun_out/r3a
cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
# the encoder with ONE exchange per step (csrc/encoder_persist1.hip): layer probe (us per step, error against the step kernels), then the bench
timeout 300 python tools/probe_persist.py 256 16 800 > $O/probe_enc.txt 2>&1; head -20 $O/probe_enc.txt
# (clusters of 8 need all 256 work-groups co-resident: skip their bench if the probe saw a cluster give up)
OHS="2 1"; grep -q "P=8.*FAILED" $O/probe_enc.txt && OHS="1"
for oh in $OHS; do
  LVSR_PERSIST_ONEHOP=$oh LVSR_PERSIST_FLAGS=64 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_onehop${oh}_staged.json 2> $O/bench_onehop${oh}_staged.err; cat $O/bench_onehop${oh}_staged.json
  LVSR_PERSIST_ONEHOP=$oh timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "encoder_forward_backward" > $O/parity_onehop$oh.txt 2>&1; tail -3 $O/parity_onehop$oh.txt
  LVSR_PERSIST_ONEHOP=$oh timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_onehop$oh.json 2> $O/bench_onehop$oh.err; cat $O/bench_onehop$oh.json
done
# the persistent decoder backward, both placements of the handler state
timeout 300 python tools/probe_decoder_persist_bwd.py wsj_base > $O/probe_bwd.txt 2>&1; tail -60 $O/probe_bwd.txt
for ls in 0 1; do
  LVSR_TEST_PBWD=1 LVSR_PBWD_LDS_STATE=$ls timeout 200 python -m pytest tests/test_gpu_properties.py -q -x -k "persistent_decoder" > $O/parity_ls$ls.txt 2>&1; tail -3 $O/parity_ls$ls.txt
  LVSR_DEC_BWD_PERSISTENT=1 LVSR_PBWD_LDS_STATE=$ls timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ls$ls.json 2> $O/bench_ls$ls.err; cat $O/bench_ls$ls.json
done
