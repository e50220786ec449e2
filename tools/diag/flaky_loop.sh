# usage: bash tools/diag/flaky_loop.sh <reps> : the first part of tests/test_gpu_parity.py over and over, failures with their stderr lines
for i in $(seq 1 $1); do
  python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "convert or quantize_clusters or kmeans_bit or many_empty or fixed_point or beyond_4096 or 5000_colours or pruned_assignment or u8_batch or sharded_batch" 2>&1 | grep -i "patolette_amd:\|patolette:\|passed\|failed" | tail -4
done
