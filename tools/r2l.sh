mkdir -p gpurun_out/r2l; O=gpurun_out/r2l
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json
d=json.load(open('$O/bench.json')); print('bench', d['ms_per_step'], d['value'], d['roofline']['us_per_recurrent_step'])"; tail -2 $O/bench.err
timeout 300 python tools/probe_persist.py 2>&1 | grep -E "steps\+graph|persist  |persist rows=2|forward input" | head -12
