mkdir -p gpurun_out/r2h; O=gpurun_out/r2h
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o dec -- python tools/bench_decode.py --utts 4 > $O/dec.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/decode_kernel_stats.md > /dev/null
head -40 $O/decode_kernel_stats.md
rm -rf $O/prof
