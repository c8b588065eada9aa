# Round 3: decode concurrency — hardware queues (GPU_MAX_HW_QUEUES) x searches in flight
cd $GRAFT_REPO_ROOT
for q in 4 8 16; do for s in 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --workload wsj_decode --utterances 64 --streams $s 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hw queues $q, in flight $s: %.2f ms per utterance, %.1f us per position' % (d['ms_per_step'], d['config']['us_per_position']))"
done; done
