#!/bin/bash
# Dev tool: SQ / TCP counters of the GPTQ mat-vec at the KAT sizes (tools/gptq_prof.py).  Usage: gptq_pmc.sh <tag> [knob2]
TAG=${1:-g}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/gptq_prof.py ${2:-0}"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/${TAG}_pmc_sq -o ${TAG} -- $CMD > $OUT/${TAG}_pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/${TAG}_pmc_sq2 -o ${TAG} -- $CMD > $OUT/${TAG}_pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o ${TAG} -- $CMD > $OUT/${TAG}_trace.log 2>&1
python - <<PY
import csv, glob, collections
for sub in ("pmc_sq", "pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/${TAG}_%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gptq" not in k: continue
            key = (k.split("(")[0].replace("void sbq::(anonymous namespace)::", "")[:60], r["Grid_Size"])
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key in sorted(acc):
        print(key, {c: round(sum(v) / len(v)) for c, v in acc[key].items()})
f = glob.glob("$OUT/${TAG}_trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if "gptq" in r["Name"]: print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
