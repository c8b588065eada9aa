mkdir -p gpurun_out/r2i; O=gpurun_out/r2i
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
timeout 300 python tools/bench_decode.py --utts 8 2>/dev/null
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o dec -- python tools/bench_decode.py --utts 4 > $O/dec.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/decode_kernel_stats.md > /dev/null
head -22 $O/decode_kernel_stats.md
rm -rf $O/prof
