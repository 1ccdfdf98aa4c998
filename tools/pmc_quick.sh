#!/bin/bash
# usage: pmc_quick.sh <tag> [layer] [kernel-name substring]   (run on the GPU box from repo root)
# HBM fetch / write and L2 hit counters of one bench layer's hot kernel, means per launch.
# FETCH_SIZE / WRITE_SIZE print in the counters' own units; tools/summarize_profile.py applies the
# gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md for the committed summaries.
TAG=$1
LAYER=${2:-conv3x3_256}
KERNEL=${3:-wino_f2_fused}
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --layer $LAYER --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>> $OUT/err.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --layer $LAYER --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>> $OUT/err.log
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python bench.py --layer $LAYER --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>> $OUT/err.log
python - <<PY
import csv,glob,collections
for kind in ("fetch","write","l2"):
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*_counter_collection.csv"%kind, recursive=True):
        for r in csv.DictReader(open(f)):
            if "$KERNEL" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(kind,k,sum(v)/len(v), len(v))
PY
