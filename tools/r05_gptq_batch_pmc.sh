#!/bin/bash
# rocprofv3 passes of the batched GPTQ mat-mul (tools/lab/gptq_batch_prof.py): kernel trace; FETCH_SIZE; WRITE_SIZE; the
# SQ counters that say what the waves did (separate --pmc passes, no tracing with them) -> gpurun_out/<tag>_gptq_batch_pmc.txt
TAG=${1:-r05gb}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/lab/gptq_batch_prof.py"
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -o ${TAG} -- $CMD > $OUT/${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o ${TAG} -- $CMD > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o ${TAG} -- $CMD > $OUT/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_sq -o ${TAG} -- $CMD > $OUT/${TAG}_sq.log 2>&1
python - <<PY > $OUT/${TAG}_gptq_batch_pmc.txt
import csv, glob, collections
def short(n):
    return n.replace("void sbq::(anonymous namespace)::", "").split("(")[0]
print("# gptq_mfma_kernel under rocprofv3 (tools/r05_gptq_batch_pmc.sh): per (kernel, grid) -- the grid identifies shape and batch tile")
print("# FETCH_SIZE / WRITE_SIZE in KB as reported; read bytes = 2 x FETCH_SIZE x 1024 on gfx950 (MI355X_MICROARCH.md), write = WRITE_SIZE x 1024")
rows = collections.defaultdict(dict)
for f in glob.glob("$OUT/${TAG}_trace/**/*kernel_trace.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gptq_mfma" not in r["Kernel_Name"]: continue
        key = (short(r["Kernel_Name"]), r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y", ""))
        acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in acc.items():
        v = sorted(v)
        rows[k]["launches"] = len(v); rows[k]["median_ns"] = v[len(v) // 2]; rows[k]["avg_ns"] = sum(v) / len(v)
for which in ("fetch", "write", "sq"):
    for f in glob.glob("$OUT/${TAG}_%s/**/*counter_collection.csv" % which, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gptq_mfma" not in r["Kernel_Name"]: continue
            key = (short(r["Kernel_Name"]), r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y", ""))
            acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            rows[k][c] = sum(v) / len(v)
for k in sorted(rows):
    print(k)
    for c in sorted(rows[k]):
        print("    %-28s %s" % (c, ("%.1f" % rows[k][c]) if isinstance(rows[k][c], float) else rows[k][c]))
PY
cat $OUT/${TAG}_gptq_batch_pmc.txt
