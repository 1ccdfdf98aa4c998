#!/bin/bash
# Run a list of GPU steps one after the other on the GPU box, each under its own timeout, logging to
# gpurun_out/<tag>/<name>.log.  A step that fails an assertion does not stop the list; a step that TIMES OUT or
# is killed does (no further GPU work after a hang).
#   tools/gpu_steps.sh <tag> "name|timeout_s|command" ...
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (timeout $tmo s): $cmd" | tee -a $OUT/steps.log
  timeout -k 10 $tmo bash -c "$cmd" > $OUT/$name.log 2>&1
  rc=$?
  echo "== $name rc=$rc" | tee -a $OUT/steps.log
  tail -3 $OUT/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -ge 128 ]; then echo "step $name timed out or was killed: stopping" | tee -a $OUT/steps.log; exit 1; fi
done
exit 0
