#!/bin/bash
# Smoke over the whole bench surface (run on the GPU box): every --layer at batches 1, 7, 16 and 128, one short trial each;
# prints layer, batch, us per layer and the executed-MFMA fraction, or FAILED.
for L in conv3x3_256 conv3x3_128 conv1x1_512_128 conv1x1_128_512 conv1x1_1024_256 conv1x1_256_1024 residual_block conv3x3_256_f4compat; do
  for B in 1 7 16 128; do
    python bench.py --layer $L --batch $B --no-cpu-baseline --steps 20 --warmup 3 --trials 1 --preheat-ms 50 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$L', $B, j['us_per_layer'], j['roofline'].get('frac'))" || echo "$L $B FAILED"
  done
done
