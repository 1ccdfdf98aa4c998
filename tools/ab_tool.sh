#!/bin/bash
# interleaved A/B of two builds of tools/ablate_fused (quick mode): tools/ab_tool.sh <old binary> <new binary> C G [reps]
OLD=$1; NEW=$2; C=${3:-256}; G=${4:-256}; R=${5:-4}
for i in $(seq $R); do
  $OLD $C $G q | sed 's/^/old: /'
  $NEW $C $G q | sed 's/^/new: /'
done
