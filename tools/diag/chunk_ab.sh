# Split loop issued chunk-major (PAMD_LQ_CHUNK_MB, pipeline.hip) against kernel-major (0), host-driven loop, one process per line.
# usage: bash tools/diag/chunk_ab.sh > gpurun_out/<dir>/chunk_ab.txt
for cfg in c3 c4; do
for mb in 0 256 128 64 0; do
PAMD_LQ_DEVICE=0 PAMD_LQ_CHUNK_MB=$mb python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('stages_ms_last_step'); k=d.get('kernels') or {}
ks=' '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('k_minmax','k_hist_lq','k_scatter_cov','k_count','k_cut','k_scan') if n in k)
print('$cfg chunk_mb=$mb ms_per_step', d['ms_per_step'], 'lq', s['ms_lq'], 'rounds', d['run']['lq_rounds'], 'evals', d['run']['split_evals'], '| kernel ms/step:', ks)"
done; done
