#!/bin/bash
# tools/graph_gap.sh <tag>: kernel-trace the block eager vs graph-replayed, print the gap analysis (GPU box)
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT/gg_trace
rocprofv3 --kernel-trace --output-format csv -d $OUT/gg_trace -- python3 tools/graph_gap.py > /dev/null 2> $OUT/gg_err.log || exit 1
python3 tools/graph_gap.py $OUT/gg_trace | tee $OUT/graph_gap.txt
rm -rf $OUT/gg_trace
