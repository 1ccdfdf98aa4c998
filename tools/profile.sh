#!/bin/bash
# Collect the rocprofv3 evidence bench.py's numbers are judged against (run on the GPU box):
#   tools/profile.sh <tag>      -> gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_sq}/...
# Counters go in their own passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2) and never together with
# sys/hip/hsa tracing.  tools/summarize_profile.py turns the CSVs into profiles/<tag>/*.
set -e
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for layer in conv3x3_256 conv3x3_128 conv1x1_512_128 conv1x1_128_512 conv1x1_1024_256 conv1x1_256_1024; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$layer -- python bench.py --layer $layer --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_trace_$layer.json 2>> $OUT/err.log
done
for layer in conv3x3_256 conv1x1_1024_256; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$layer -- python bench.py --layer $layer --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>> $OUT/err.log
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$layer -- python bench.py --layer $layer --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>> $OUT/err.log
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq_$layer -- python bench.py --layer $layer --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>> $OUT/err.log
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_l2_$layer -- python bench.py --layer $layer --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>> $OUT/err.log
done
python bench.py --steps 200 > $OUT/bench_unprofiled.json 2>> $OUT/err.log
echo done
