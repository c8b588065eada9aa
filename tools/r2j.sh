mkdir -p gpurun_out/r2j; O=gpurun_out/r2j
timeout 1200 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | head -c 600; echo; tail -2 $O/bench.err
timeout 900 python bench.py --workload wsj_decode --utterances 200 > $O/bench_decode.json 2> $O/bench_decode.err; cat $O/bench_decode.json; tail -2 $O/bench_decode.err
