# Round 3: the plain-store hand-off (persist.h cluster_shares_xcd) in the encoder and decoder cluster kernels.
#   gpurun --timeout 900 -- 'bash tools/r3b.sh'
mkdir -p gpurun_out/r3b; O=gpurun_out/r3b
cd $GRAFT_REPO_ROOT
timeout 300 python tools/probe_persist.py 256 16 800 512 8 800 > $O/probe_enc.txt 2>&1; cat $O/probe_enc.txt
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
LVSR_DEC_BWD_PERSISTENT=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_pbwd.json 2> $O/bench_pbwd.err; cat $O/bench_pbwd.json
LVSR_PERSIST_FLAGS=4 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_sc1.json 2> $O/bench_sc1.err; cat $O/bench_sc1.json
timeout 300 python tools/probe_decoder_persist.py wsj_base > $O/probe_dec.txt 2>&1; tail -30 $O/probe_dec.txt
timeout 300 python tools/probe_decoder_persist_bwd.py wsj_base > $O/probe_bwd.txt 2>&1; tail -40 $O/probe_bwd.txt
LVSR_TEST_PBWD=1 timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
