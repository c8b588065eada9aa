# persistent decoder on the other workloads / launch modes
for w in timit_tiny; do for p in 0 auto; do LVSR_DEC_PERSISTENT=$p timeout 200 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w persistent=$p', round(d['ms_per_step'],3), round(d['value']))"; done; done
timeout 200 python bench.py --ragged --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ragged', round(d['ms_per_step'],3), round(d['value']))"
timeout 300 python bench.py --force-dist --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('one RCCL rank', round(d['ms_per_step'],3), round(d['value']))"
timeout 300 python bench.py --batch 32 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch 32', round(d['ms_per_step'],3), round(d['value']))"
