for cfg in c2 c3 c4; do
for mode in 1 0; do
PAMD_LQ_DEVICE=$mode python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('stages_ms_last_step')
print('$cfg device_loop=$mode ms_per_step', d['ms_per_step'], 'gq', s['ms_gq'], 'lq', s['ms_lq'], 'rounds', d['run']['lq_rounds'], 'evals', d['run']['split_evals'])"
done; done
