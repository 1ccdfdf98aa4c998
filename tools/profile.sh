#!/bin/bash
# Collect the rocprofv3 evidence bench.py's numbers are judged against (run on the GPU box):
#   tools/profile.sh <tag> [configs...]   -> gpurun_out/prof_<tag>/<config>/{trace,fetch,write,sq,l2}/...
# One kernel-trace pass (--stats) and four counter passes per config.  Counters go in their own
# passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2) and never together with sys/hip/hsa tracing; the
# program after `--` is python3 itself (no env/bash hop).  tools/summarize_profile.py turns the CSVs
# into profiles/<tag>/.  Configs = every BASELINE.json config plus the residual block, the N = 1
# latency kernel and the F(4x4) compatibility path (whose stand-alone transform stages are the
# HBM-bound kernels north_star asks GB/s for).
TAG=${1:-r3}
[ $# -gt 0 ] && shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CONFIGS=${@:-"conv3x3_256 conv3x3_128 conv1x1_512_128 conv1x1_128_512 conv1x1_1024_256 conv1x1_256_1024 residual_block conv3x3_256_f4compat conv3x3_256@1 conv3x3_128@1 conv1x1_1024_256@1 conv1x1_512_128@1 conv1x1_128_512@1 conv1x1_256_1024@1 conv3x3_256@16"}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT"
for cfg in $CONFIGS; do
  layer=${cfg%@*}
  batch=128
  [[ "$cfg" == *@* ]] && batch=${cfg#*@}
  D=$OUT/$cfg
  mkdir -p $D
  ARGS="bench.py --layer $layer --batch $batch --no-cpu-baseline --trials 1"
  rm -f $D/FAILED
  # one pass = one rocprofv3 run of python3 itself; a pass that fails marks the config (summarize_profile.py
  # skips marked configs) instead of leaving empty or stale CSVs behind
  pass() {   # pass <subdir> <stdout file> <rocprofv3 options...> -- <bench args...>
    local sub=$1 so=$2; shift 2
    rm -rf $D/$sub
    if ! rocprofv3 --kernel-trace "$@" > $so 2>> $OUT/err.log; then echo "$sub" >> $D/FAILED; fi
  }
  pass trace $D/bench_trace.json --stats --output-format csv -d $D/trace -- python3 $ARGS --steps 50 --warmup 5
  pass fetch /dev/null --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $ARGS --steps 10 --warmup 2 --preheat-ms 50
  pass write /dev/null --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $ARGS --steps 10 --warmup 2 --preheat-ms 50
  pass sq /dev/null --pmc $SQ --output-format csv -d $D/sq -- python3 $ARGS --steps 10 --warmup 2 --preheat-ms 50
  pass l2 /dev/null --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $D/l2 -- python3 $ARGS --steps 10 --warmup 2 --preheat-ms 50
  if ! python3 bench.py --layer $layer --batch $batch --no-cpu-baseline --steps 200 > $D/bench_unprofiled.json 2>> $OUT/err.log; then echo unprofiled >> $D/FAILED; fi
  if [ -f $D/FAILED ]; then echo "FAILED $cfg: $(tr '\n' ' ' < $D/FAILED)"; else echo "profiled $cfg"; fi
done
echo done
