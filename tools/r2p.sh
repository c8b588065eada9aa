# which block type of attbwd_post costs what: bench with parts ablated (results wrong, timing only)
for m in 0 1 2 4 7; do LVSR_ATTBWD_POST_ABLATE=$m timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ablate $m', round(d['ms_per_step'],3))"; done
