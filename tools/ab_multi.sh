#!/bin/bash
# interleaved quick-mode runs of several builds of tools/ablate_fused: tools/ab_multi.sh C G reps bin1 bin2 ...
C=$1; G=$2; R=$3; shift 3
for i in $(seq $R); do for b in "$@"; do $b $C $G q | sed "s|^|$(basename $b): |" | cut -c1-170; done; done
