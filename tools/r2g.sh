mkdir -p gpurun_out/r2g; O=gpurun_out/r2g
timeout 600 python -m pytest tests -m gpu -q -x -k "beam or topk or lm or decode or persistent" > $O/gpu_beam_tests.log 2>&1; tail -8 $O/gpu_beam_tests.log
timeout 300 python tools/bench_decode.py --utts 8 > $O/decode8.json 2> $O/decode8.err; cat $O/decode8.json; tail -3 $O/decode8.err
timeout 300 python tools/bench_decode.py --utts 8 --no-lm 2>/dev/null
LVSR_STEP_GRAPH=0 timeout 300 python tools/bench_decode.py --utts 8 2>/dev/null
timeout 600 python bench.py --workload wsj_decode --utterances 40 > $O/bench_decode.json 2> $O/bench_decode.err; cat $O/bench_decode.json; tail -3 $O/bench_decode.err
