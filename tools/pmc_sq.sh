#!/bin/bash
# Developer tool (run on the GPU box from the repo root): where do the fused kernel's wave cycles go?
# Three rocprofv3 --pmc passes over bench.py's headline layer; prints per-launch means.
TAG=${1:-sq}
OUT=gpurun_out/pmcsq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
P3="SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA"
P4="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python bench.py --layer conv3x3_256 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>> $OUT/err.log
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wino_f2_fused" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
wc = sum(agg["SQ_WAVE_CYCLES"])/max(1,len(agg["SQ_WAVE_CYCLES"]))
for k in sorted(agg):
    m=sum(agg[k])/len(agg[k]); print("%-32s %14.0f  %6.3f of wave cycles" % (k, m, m/wc if wc else 0))
PY
