#!/bin/bash
# One gpurun call of the round: GPU tests, then the measurements named on the command line, everything under its own timeout,
# logs under gpurun_out/<tag>/.   usage: gpurun -- 'bash tools/gpu_session.sh <tag> [tests] [bench] [dec] [deep] [long] [stack] [gemm] ...'
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
export PYTHONUNBUFFERED=1
B="--no-cpu-baseline --no-decode --no-fbank --no-strong --no-ragged --sustained-seconds 0"
for what in "$@"; do
  case $what in
    tests) timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep "^FAILED\|^ERROR" $O/pytest.log | cut -c1-200 | head -n 30; tail -n 4 $O/pytest.log;;
    fbank) timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-strong --no-ragged --sustained-seconds 0 > $O/fbank.json 2> $O/fbank.err; python -c "import json;d=json.load(open('$O/fbank.json'));print({k:(v.get('achieved'),v.get('frac'),v.get('launch_us')) for k,v in d.get('fbank',{}).items() if isinstance(v,dict)})"; tail -n 2 $O/fbank.err;;
    k:*) timeout 900 python -m pytest tests -m gpu -x -q -k "${what#k:}" > "$O/pytest_k_$(echo ${what#k:} | cut -c1-12 | tr " " _).log" 2>&1; echo "pytest -k rc=$?"; grep "^E " $O/pytest_k_*.log | cut -c1-300 | head -n 12; tail -n 3 $O/pytest_k_*.log;;
    testsk:*) k=${what#testsk:}; timeout 900 python -m pytest tests -m gpu -x -q --knob $k > $O/pytest_$k.log 2>&1; echo "pytest --knob $k rc=$?"; tail -n 3 $O/pytest_$k.log;;
    newtests) timeout 900 python -m pytest tests -m gpu -x -q -k "wsj_base_median or whole_list or persistent_decoder or wsj_deep or wsj_paper or stack2" > $O/pytest_new.log 2>&1; echo "pytest(new) rc=$?"; tail -n 6 $O/pytest_new.log;;
    propparity) timeout 900 python tools/full_size_parity.py prop_median prop_mean > $O/prop_parity.md 2> $O/prop_parity.err; echo "propparity rc=$?"; cat $O/prop_parity.md | cut -c1-330; tail -n 3 $O/prop_parity.err;;
    forcedist) timeout 300 python bench.py --steps 10 --warmup 3 --force-dist $B > $O/forcedist.json 2> $O/forcedist.err; echo "forcedist rc=$?"; python -c "import json;d=json.load(open('$O/forcedist.json'));print('force-dist', d['ms_per_step'], d['value'], d['self_check'], {k:d['config'].get(k) for k in ('collective_backend','collective_world_size','allreduce_ms','whole_step_graph_region')})"; tail -n 2 $O/forcedist.err;;
    parity) timeout 1500 python tools/full_size_parity.py > $O/full_size_parity.md 2> $O/full_size_parity.err; echo "parity rc=$?"; cat $O/full_size_parity.md | cut -c1-330; tail -n 3 $O/full_size_parity.err;;
    nk:*) timeout 1500 python -m pytest tests -m gpu -q -k "${what#nk:}" > "$O/pytest_nk.log" 2>&1; echo "pytest -k rc=$?"; grep "^E \|^FAILED\|^ERROR" $O/pytest_nk.log | cut -c1-400 | head -n 40; tail -n 3 $O/pytest_nk.log;;
    ragged) timeout 300 python bench.py --steps 10 --warmup 3 --ragged $B > $O/ragged.json 2> $O/ragged.err; python -c "import json;d=json.load(open('$O/ragged.json'));print('ragged', d['ms_per_step'], d['value'])"; tail -n 1 $O/ragged.err;;
    bench) timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "import json;d=json.load(open('$O/bench.json'));print({k:d.get(k) for k in ('value','ms_per_step','self_check','ragged')}, d.get('decode'), d.get('sustained'), d.get('strong'), {k:(v.get('achieved'),v.get('frac'),v.get('launch_us')) for k,v in d.get('fbank',{}).items() if isinstance(v,dict)}, d['roofline']['us_per_recurrent_step'], d['roofline']['dense_gemm']['layer_shapes'])" ; tail -n 3 $O/bench.err;;
    quick) timeout 300 python bench.py --steps 20 --warmup 5 $B > $O/quick.json 2> $O/quick.err; echo "quick rc=$?"; python -c "import json;d=json.load(open('$O/quick.json'));print('wsj_base', d['ms_per_step'], d['value'], d['roofline']['us_per_recurrent_step'])"; tail -n 2 $O/quick.err;;
    nodrain) timeout 300 python bench.py --steps 40 --warmup 5 --no-drain $B --no-ragged > $O/nodrain.json 2> $O/nodrain.err; python -c "import json;d=json.load(open('$O/nodrain.json'));print('no-drain', d['ms_per_step'], d['value'], d['self_check'])"; tail -n 2 $O/nodrain.err
             timeout 300 python bench.py --steps 40 --warmup 5 $B --no-ragged > $O/drain.json 2> $O/drain.err; python -c "import json;d=json.load(open('$O/drain.json'));print('drain', d['ms_per_step'], d['value'])";;
    quick8) timeout 300 python bench.py --steps 20 --warmup 5 $B --knob dec_cluster=8 > $O/quick8.json 2> $O/quick8.err; python -c "import json;d=json.load(open('$O/quick8.json'));print('wsj_base clusters of 8', d['ms_per_step'], d['value'])"; tail -n 2 $O/quick8.err;;
    dec) for k in dec_cluster=0 dec_cluster=8; do timeout 300 python tools/probe_decoder_persist.py wsj_base $k > $O/dec_fwd_$k.txt 2>&1; timeout 300 python tools/probe_decoder_persist_bwd.py wsj_base $k > $O/dec_bwd_$k.txt 2>&1; echo "== $k"; grep -v "^    " $O/dec_fwd_$k.txt | tail -n 4; grep -v "^    " $O/dec_bwd_$k.txt | tail -n 4; done;;
    decab:*) k=${what#decab:}; for kk in persist_flags=0 $k persist_flags=0 $k; do timeout 300 python tools/probe_decoder_persist_bwd.py wsj_base $kk > $O/decab_$kk.txt 2>&1; echo "== $kk"; grep "persistent:\|energies\|D gather\|A gather\|publish dpc\|sum" $O/decab_$kk.txt; done;;
    skew) timeout 600 python tools/probe_decoder_bwd_skew.py wsj_base > $O/dec_bwd_skew.txt 2>&1; grep -v amdgpu.ids $O/dec_bwd_skew.txt | tail -n 20;;
    decmed) timeout 300 python tools/probe_decoder_persist.py wsj_base median > $O/dec_fwd_median.txt 2>&1; grep -v "^    " $O/dec_fwd_median.txt | tail -n 4;;
    deep) timeout 600 python bench.py --workload wsj_deep --steps 5 --warmup 2 $B > $O/deep.json 2> $O/deep.err; echo "deep rc=$?"; python -c "import json;d=json.load(open('$O/deep.json'));print('wsj_deep', d['ms_per_step'], d['value'])"; tail -n 2 $O/deep.err
          timeout 300 python tools/probe_decoder_persist.py wsj_deep > $O/deep_dec_fwd.txt 2>&1; timeout 300 python tools/probe_decoder_persist_bwd.py wsj_deep > $O/deep_dec_bwd.txt 2>&1; grep -v "^    " $O/deep_dec_fwd.txt | tail -n 4; grep -v "^    " $O/deep_dec_bwd.txt | tail -n 4;;
    long) for T in 864 1000 1200 1600; do timeout 300 python bench.py --steps 10 --warmup 3 --frames $T $B > $O/frames_$T.json 2> $O/frames_$T.err; python -c "import json;d=json.load(open('$O/frames_$T.json'));print('frames $T', d['ms_per_step'], d['value'])"; tail -n 1 $O/frames_$T.err; done;;
    stack) timeout 300 python bench.py --workload wsj_stack2 --steps 10 --warmup 3 $B > $O/stack2.json 2> $O/stack2.err; python -c "import json;d=json.load(open('$O/stack2.json'));print('wsj_stack2', d['ms_per_step'], d['value'])"; tail -n 1 $O/stack2.err;;
    median) timeout 300 python bench.py --workload wsj_base_median --steps 10 --warmup 3 $B > $O/median.json 2> $O/median.err; python -c "import json;d=json.load(open('$O/median.json'));print('wsj_base_median', d['ms_per_step'], d['value'])"; tail -n 1 $O/median.err;;
    paper) timeout 300 python bench.py --workload wsj_paper --steps 10 --warmup 3 $B > $O/paper.json 2> $O/paper.err; python -c "import json;d=json.load(open('$O/paper.json'));print('wsj_paper', d['ms_per_step'], d['value'])"; tail -n 1 $O/paper.err;;
    batches) for b in 10 32 64 128; do timeout 300 python bench.py --steps 8 --warmup 2 --batch $b $B > $O/batch_$b.json 2> $O/batch_$b.err; python -c "import json;d=json.load(open('$O/batch_$b.json'));print('batch $b', d['ms_per_step'], d['value'])"; tail -n 1 $O/batch_$b.err; done;;
    gemm) timeout 400 python tools/probes/gemm_k_sweep.py sustained > $O/gemm_k_sweep.txt 2>&1; tail -n 22 $O/gemm_k_sweep.txt;;
    prof) cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --steps 10 --warmup 3 $B > $R/$O/prof.log 2>&1; cd $R; ls $O/prof | head; python tools/rocpd_stats.py $O/prof/*/*.db > $O/kernel_stats.md 2>> $O/prof.log || python tools/rocpd_stats.py $O/prof/*.db > $O/kernel_stats.md 2>> $O/prof.log; python tools/rocpd_timeline.py $(find $O/prof -name "*.db" | head -n 1) --list > $O/timeline.txt 2>> $O/prof.log; head -n 30 $O/kernel_stats.md; head -n 25 $O/timeline.txt; rm -rf $O/prof;;
    pmc) cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
         for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY"; do
           n=$(echo $c | cut -d" " -f1)
           timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$n -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-graph $B > $R/$O/pmc_$n.log 2>&1; echo "pmc $n rc=$?"
         done
         cd $R; python tools/pmc_summary.py --json $O/pmc_bench.json --tag wsj_base $(find $O/pmc_* -name "*.db") > $O/pmc_bench_wsj_base.md 2>> $O/pmc_FETCH_SIZE.log; head -n 12 $O/pmc_bench_wsj_base.md | cut -c1-260; python -c "import json;d=json.load(open('$O/pmc_bench.json'));print({k:v for k,v in d.items() if 'enc_p' in k or k=='__stamp__'})";;
    fbpmc) cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
         for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
           n=$(echo $c | cut -d" " -f1)
           timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/$O/fbpmc_$n -o pmc -- python $R/tools/probes/fbank_batch_probe.py > $R/$O/fbpmc_$n.log 2>&1; echo "fbpmc $n rc=$?"
         done
         cd $R; python tools/pmc_summary.py --json $O/fbpmc.json --tag fbank $(find $O/fbpmc_* -name "*.db") > $O/pmc_fbank.md 2>> $O/fbpmc_FETCH_SIZE.log; grep -i "fbank\|deltas\|kernel |" $O/pmc_fbank.md | cut -c1-300;;
    decpmc) cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
         for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY"; do
           n=$(echo $c | cut -d" " -f1)
           timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/$O/decpmc_$n -o pmc -- python $R/tools/bench_decode.py --utts 32 --batch 32 --streams 1 > $R/$O/decpmc_$n.log 2>&1; echo "decpmc $n rc=$?"
         done
         cd $R; python tools/pmc_summary.py --json $O/decpmc.json --tag decode $(find $O/decpmc_* -name "*.db") > $O/pmc_decode.md 2>> $O/decpmc_FETCH_SIZE.log; grep "attdec\|beam\|readout\|fst\|kernel |" $O/pmc_decode.md | cut -c1-260;;
    timeline) python tools/rocpd_timeline.py $(find $O/prof -name "*.db" | head -n 1) > $O/timeline.txt 2>&1; head -n 24 $O/timeline.txt;;
    dectests) timeout 900 python -m pytest tests -m gpu -x -q -k "batched or decode or beam" > $O/pytest_dec.log 2>&1; echo "pytest(decode) rc=$?"; grep "^E " $O/pytest_dec.log | cut -c1-300 | head -n 12; tail -n 3 $O/pytest_dec.log;;
    deck:*) x=${what#deck:}; bb=${x%%:*}; k=${x#*:}; timeout 600 python tools/bench_decode.py --utts 128 --batch ${bb%x*} --streams ${bb#*x} --knob $k > $O/deck_${bb}_$k.json 2> $O/deck_${bb}_$k.err; echo "$k: $(cut -c1-330 $O/deck_${bb}_$k.json)"; tail -n 1 $O/deck_${bb}_$k.err;;
    decb:*) bb=${what#decb:}; timeout 600 python tools/bench_decode.py --utts 128 --batch ${bb%x*} --streams ${bb#*x} > $O/decb_$bb.json 2> $O/decb_$bb.err; cat $O/decb_$bb.json; tail -n 2 $O/decb_$bb.err;;
    decbprof:*) bb=${what#decbprof:}; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
         timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/decbprof -o dec -- python $R/tools/bench_decode.py --utts 64 --batch ${bb%x*} --streams ${bb#*x} > $R/$O/decbprof.log 2>&1
         cd $R; python tools/rocpd_stats.py $(find $O/decbprof -name "*.db" | head -n 1) > $O/decode_batched_kernel_stats.md 2>&1; tail -n 2 $O/decbprof.log | cut -c1-400; head -n 40 $O/decode_batched_kernel_stats.md | cut -c1-160;;
    decodeb:*) bb=${what#decodeb:}; timeout 900 python bench.py --workload wsj_decode --decode-batch ${bb%x*} --streams ${bb#*x} > $O/decode_$bb.json 2> $O/decode_$bb.err; python -c "import json;d=json.load(open('$O/decode_$bb.json'));print('wsj_decode $bb', d['ms_per_step'], d['value'])"; tail -n 1 $O/decode_$bb.err;;
    decode) timeout 900 python bench.py --workload wsj_decode > $O/decode.json 2> $O/decode.err; echo "decode rc=$?"; cut -c1-600 $O/decode.json; tail -n 2 $O/decode.err;;
    smoke) timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log;;
    beam200) timeout 400 python tools/bench_decode.py --beam 200 --conditioned --utts 32 --batch 8 --streams 2 > $O/b200.json 2> $O/b200.err; python -c "import json;d=json.loads(open('$O/b200.json').read().strip().split('\n')[-1]);print('beam200', round(d['sec_per_utt']*1e3,2), 'ms/utt', round(d['us_per_position'],1), 'us/position')";;
    beam200k) for er in 0 16 32 64 200; do timeout 200 python tools/bench_decode.py --beam 200 --utts 32 --batch 8 --streams 2 --knob energy_rows=$er > $O/b200_$er.json 2> $O/b200_$er.err; python -c "import json;d=json.loads(open('$O/b200_$er.json').read().strip().split('\n')[-1]);print('beam200 energy_rows=$er', round(d['sec_per_utt']*1e3,2), 'ms/utt', round(d['us_per_position'],1), 'us/position')"; done;;
    prof200) cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
         timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof200 -o dec -- python $R/tools/bench_decode.py --utts 16 --beam 200 --conditioned --batch 8 --streams 1 > $R/$O/prof200.log 2>&1
         cd $R; python tools/rocpd_stats.py $(find $O/prof200 -name "*.db" | head -n 1) > $O/decode200_kernel_stats.md 2>&1; tail -n 2 $O/prof200.log | cut -c1-400; head -n 30 $O/decode200_kernel_stats.md | cut -c1-160; rm -rf $O/prof200;;
    decprof) cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
         for st in 1 8; do timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/decprof$st -o dec -- python $R/tools/bench_decode.py --utts 8 --streams $st > $R/$O/decprof$st.log 2>&1; done
         cd $R; for st in 1 8; do python tools/rocpd_stats.py $(find $O/decprof$st -name "*.db" | head -n 1) > $O/decode_kernel_stats_$st.md 2>&1; tail -n 2 $O/decprof$st.log | cut -c1-400; head -n 34 $O/decode_kernel_stats_$st.md | cut -c1-150; done;;
    profb:*) bb=${what#profb:}; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/profb$bb -o bench -- python $R/bench.py --steps 6 --warmup 2 --batch $bb $B > $R/$O/profb$bb.log 2>&1; cd $R; python tools/rocpd_timeline.py $(find $O/profb$bb -name "*.db" | head -n 1) > $O/timeline_b$bb.txt 2>&1; head -n 12 $O/timeline_b$bb.txt; python tools/rocpd_stats.py $(find $O/profb$bb -name "*.db" | head -n 1) > $O/kernel_stats_b$bb.md 2>&1; head -n 9 $O/kernel_stats_b$bb.md | cut -c1-140;;
    fbtests) timeout 300 python -m pytest tests/test_fbank.py -m gpu -x -q > $O/pytest_fb.log 2>&1; echo "pytest(fbank) rc=$?"; tail -n 3 $O/pytest_fb.log;;
    enctests) timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "encoder or full_size" > $O/pytest_enc.log 2>&1; echo "pytest(enc) rc=$?"; tail -n 4 $O/pytest_enc.log;;
    bk:*) x=${what#bk:}; bb=${x%%:*}; k=${x#*:}; timeout 300 python bench.py --steps 8 --warmup 2 --batch $bb $B --knob $k > $O/batch_${bb}_$k.json 2> $O/batch_${bb}_$k.err; python -c "import json;d=json.load(open('$O/batch_${bb}_$k.json'));print('batch $bb $k', d['ms_per_step'], d['value'], d['roofline']['us_per_recurrent_step'])"; tail -n 1 $O/batch_${bb}_$k.err;;
    knob:*) k=${what#knob:}; timeout 300 python bench.py --steps 20 --warmup 5 $B --knob $k > $O/quick_$k.json 2> $O/quick_$k.err; python -c "import json;d=json.load(open('$O/quick_$k.json'));print('wsj_base $k', d['ms_per_step'], d['value'], d['roofline']['us_per_recurrent_step'])"; tail -n 1 $O/quick_$k.err;;
    tiny) timeout 300 python bench.py --workload timit_tiny --steps 50 --warmup 10 $B > $O/timit_tiny.json 2> $O/timit_tiny.err; python -c "import json;d=json.load(open('$O/timit_tiny.json'));print('timit_tiny', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'])"; tail -n 1 $O/timit_tiny.err;;
    ragdev) timeout 900 python tools/ragged_deviation.py wsj_base_ragged --out=$O/ragged_deviation.md > /dev/null 2> $O/ragged_deviation.err; echo "ragdev rc=$?"; grep -v "^$" $O/ragged_deviation.md | cut -c1-400 | tail -n 45; tail -n 3 $O/ragged_deviation.err;;
    large) timeout 1500 python -m pytest tests/test_gpu_properties.py -m gpu -q -k "large_per_gpu or in_passes" > $O/pytest_large.log 2>&1; echo "pytest(large) rc=$?"; grep "^E  \|^FAILED" $O/pytest_large.log | cut -c1-600 | head -n 30; tail -n 3 $O/pytest_large.log;;
    *) echo "unknown item $what";;
  esac
done
