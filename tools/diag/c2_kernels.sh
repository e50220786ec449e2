# per-kernel HIP-event times of a C2 step (bench.py's kernel table), device-side global quantiser on / off
for m in 1 0; do PAMD_GQ_DEVICE=$m python bench.py --config c2 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_last_step']; k=d.get('kernels') or {}
print('c2 gq_device=$m ms_per_step', d['ms_per_step'], 'gq', s['ms_gq'], 'lq', s['ms_lq'], 'map', s['ms_map'])
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step']): print('   %-18s %.4f ms/step  %5.1f launches' % (n, v['ms_per_step'], v['launches_per_step']))"
done
