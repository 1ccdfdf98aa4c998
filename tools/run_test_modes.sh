#!/bin/bash
# The reference's own protocol on the GPU box: generate the data set (this repo's byte-identical generator),
# then ./Test 0..5 (100 calls each, first two discarded, Test.c:14,45-53); prints each mode's last two lines.
#   tools/run_test_modes.sh [N] [iters]
N=${1:-1}; IT=${2:-100}
W=$(mktemp -d /tmp/wino_test.XXXX)
python3 cuda-winograd_amd/data_generator.py --out $W/data > /dev/null || exit 1
ROOT=$(pwd)
cd $W
for m in 0 1 2 3 4 5; do
  WINO_CPU_BASELINE=0 $ROOT/Test $m $N 1 $IT | grep -E "^Average|^\{" || exit 1
done
rm -rf $W
