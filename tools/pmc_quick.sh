#!/bin/bash
# usage: pmc_quick.sh <tag>  (run on the GPU box from repo root)
TAG=$1
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --layer conv3x3_256 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>> $OUT/err.log
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python bench.py --layer conv3x3_256 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>> $OUT/err.log
python - <<PY
import csv,glob,collections
for kind in ("fetch","l2"):
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*_counter_collection.csv"%kind, recursive=True):
        for r in csv.DictReader(open(f)):
            if "wino_f2_fused" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(kind,k,sum(v)/len(v), len(v))
PY
