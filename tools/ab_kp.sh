#!/bin/bash
# A/B of the per-k-block tail in phase order (WINO_SK_KP=1, default), without the phase order (2) and round 2's
# item-major tail (WINO_SK_KP=0): time, then
# HBM-side traffic and L2 hit rate of the headline layer (GPU box).   tools/ab_kp.sh <tag>
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for rep in 1 2; do
  for kp in 0 2 1; do
    WINO_SK_KP=$kp python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --trials 5 > $OUT/bench_kp${kp}_$rep.json 2>> $OUT/err.log
    python3 - <<PY
import json
j=json.loads(open("$OUT/bench_kp${kp}_$rep.json").read().strip().splitlines()[-1])
print("kp=$kp rep=$rep  trials_us", j["trials_us"], "kernel_us", j["roofline"]["trials_kernel_us"], "clock", j["roofline"]["trials_clock_ghz"])
PY
  done
done
for kp in 0 2 1; do
  WINO_SK_KP=$kp tools/pmc_quick.sh ${TAG}_kp$kp > $OUT/pmc_kp$kp.txt 2>&1
  echo "kp=$kp"; cat $OUT/pmc_kp$kp.txt
done
