# Round 3: persistent decoder backward as default + unhoisted forward + wide (8 work-group) encoder clusters.
#   gpurun --timeout 900 -- 'bash tools/r3c.sh'
mkdir -p gpurun_out/r3c; O=gpurun_out/r3c
cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --knob persist_flags=64 --knob max_cluster_wgs=256 > $O/bench_wide.json 2> $O/bench_wide.err; cat $O/bench_wide.json; tail -3 $O/bench_wide.err
timeout 300 python tools/probe_persist.py 256 16 800 > $O/probe_enc.txt 2>&1; cat $O/probe_enc.txt
timeout 300 python tools/probe_decoder_persist.py wsj_base > $O/probe_dec.txt 2>&1; head -4 $O/probe_dec.txt
LVSR_KNOB_PERSIST_FLAGS=64 LVSR_KNOB_MAX_CLUSTER_WGS=256 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "encoder_forward_backward" > $O/parity_wide.txt 2>&1; tail -3 $O/parity_wide.txt
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
