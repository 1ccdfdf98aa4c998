#!/bin/bash
# Kernel-trace the small-batch cases (run on the GPU box): tools/trace_latency.sh <tag> [quick|full]
TAG=${1:-r3}; MODE=${2:-quick}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT/lat_trace
rocprofv3 --kernel-trace --output-format csv -d $OUT/lat_trace -- python3 tools/latency_cases.py $MODE > $OUT/lat_cases.jsonl 2> $OUT/lat_err.log || exit 1
python3 tools/trace_split.py $OUT/lat_trace $OUT/lat_cases.jsonl $OUT/latency_$MODE.json
# the raw trace is large: keep the summary only
rm -rf $OUT/lat_trace
