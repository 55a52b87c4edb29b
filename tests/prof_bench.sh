#!/bin/bash
# rocprofv3 passes over the default bench command (kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate passes)
# the library is built (if stale) BEFORE any profiled process starts: a profiled bench never spawns a compiler, all passes measure one build
(cd $GRAFT_REPO_ROOT && python3 -c "from tensegrity_rl_amd import build; build.build()" > /dev/null) || exit 1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
ARGS="--steps 100 --warmup 30 --no-cpu-baseline --no-vecenv"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/sq.json 2> $OUT/sq.err
python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
res = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows if "tgr::" in r["Name"]]
# with the donor stream one tgr_step launches step_kernel_row twice -- live rows (grid = n / 4 blocks) and donor rows (grid = donors / 4 blocks):
# the per-dispatch trace tells them apart by grid size
by_grid = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel_row" in r["Kernel_Name"]:
            by_grid[int(r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res["step_kernel_row_by_grid_threads"] = {str(g): {"calls": len(v), "avg_ns": sum(v) / len(v), "min_ns": min(v), "max_ns": max(v)} for g, v in sorted(by_grid.items())}
for tag in ("fetch", "write", "sq"):
    for f in glob.glob(out + f"/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if "tgr::" not in r["Kernel_Name"]: continue
            k = r["Kernel_Name"].split("<")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
        res[tag] = {k: {"dispatches": len(disp[k]), **{c: v for c, v in d.items()}} for k, d in acc.items()}
res["bench_line_under_trace"] = open(out + "/trace.json").read().strip()
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tensegrity_rl_amd import build as _b
res["src_hash"] = _b._src_hash()
ks = sorted(res.get("kernel_stats", []), key=lambda r: -float(r["TotalDurationNs"]))
res["dominant_kernel"] = ks[0]["Name"].split("<")[0].replace("void ", "") if ks else None
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
