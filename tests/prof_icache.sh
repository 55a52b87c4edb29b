#!/bin/bash
# instruction-cache and stall counters for the step kernel (diagnostic)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
rocprofv3 -L > $OUT/avail.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAVE_DEP[A-Z_]*\|SQC_[A-Z_]*" $OUT/avail.txt | sort -u > $OUT/names.txt
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/ic -- python3 $GRAFT_REPO_ROOT/tests/gpu_quickbench.py 4096 tensegrity_env-v0 60 1 > $OUT/ic.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/ic/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("<")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        print(k, dict(d))
PY
