B="--no-cpu-baseline --no-decode --no-fbank --no-strong --no-ragged --sustained-seconds 0"
mkdir -p gpurun_out/r09q
for rep in 1 2; do
for w in timit_tiny wsj_paper; do
  for k in persist_flags=0 persist_flags=16384 persist_flags=32768; do
    st=20; [ $w = timit_tiny ] && st=50
    timeout 300 python bench.py --workload $w --steps $st --warmup 5 $B --knob $k > gpurun_out/r09q/${w}_${k}_$rep.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/r09q/${w}_${k}_$rep.json'));print('$w $k', d['ms_per_step'])"
  done
done
done
