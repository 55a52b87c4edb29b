#!/bin/bash
# kernel-trace only (development aid): usage prof_trace.sh <tag> <lib.so> <quickbench args...>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
export TGR_LIB=$GRAFT_REPO_ROOT/$2
shift 2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tests/gpu_quickbench.py "$@" > $OUT/trace.log 2> $OUT/trace.err
grep "^n=" $OUT/trace.log | sed 's/.*stats/stats/'
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tgr::" in r["Name"]: print(r["Name"][:60], r["Calls"], "avg_us %.1f" % (float(r["AverageNs"])/1e3), "min_us %.1f max_us %.1f" % (float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
