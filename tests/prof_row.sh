#!/bin/bash
# SQ counters of the row kernels under the quick bench (development aid): usage prof_row.sh <tag> <lib.so> <env-id> [extra quickbench args]
(cd $GRAFT_REPO_ROOT && python3 -c "from tensegrity_rl_amd import build; build.build()" > /dev/null) || exit 1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
export TGR_LIB=$GRAFT_REPO_ROOT/$2
shift 2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tests/gpu_quickbench.py 4096 "$@" > $OUT/trace.log 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/tests/gpu_quickbench.py 4096 "$@" > $OUT/sq.log 2> $OUT/sq.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/tests/gpu_quickbench.py 4096 "$@" > $OUT/sq2.log 2> $OUT/sq2.err
python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
res = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")} for r in rows if "tgr::" in r["Name"]]
for tag in ("sq", "sq2"):
    for f in glob.glob(out + f"/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if "tgr::" not in r["Kernel_Name"]: continue
            k = r["Kernel_Name"].split("<")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
        res[tag] = {k: {"dispatches": len(disp[k]), **{c: v for c, v in d.items()}} for k, d in acc.items()}
res["line"] = [l for l in open(out + "/trace.log") if l.startswith("n=")]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
