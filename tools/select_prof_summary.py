"""Condense the rocprofv3 output of tools/rocprof_select.sh: per kernel calls / avg / min / max duration (kernel
trace) and FETCH_SIZE per launch (its own --pmc pass).  FETCH_SIZE is reported in KB and, on gfx950, counts each
128-byte request of a wide coalesced stream as 64 bytes (MI355X_MICROARCH.md, HBM section; tools/pmc_summary.py uses
the same correction): read bytes = 2 x FETCH_SIZE x 1024."""
import csv, glob, os, sys, collections, re

out, tag = sys.argv[1], sys.argv[2]
what = sys.argv[3] if len(sys.argv) > 3 else "sel"   # "sel": tools/rocprof_select.sh, "gptq": tools/rocprof_gptq.sh
by_grid = what == "gptq"                            # one row per (kernel, grid): the shapes differ by grid

def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("sbq::(anonymous namespace)::", "").replace("sbq::", "")
    return re.sub(r"\(.*$", "", name)

dur = collections.defaultdict(list)
def key(r, grid_field):
    k = short(r["Kernel_Name"])
    return k + " grid=" + r[grid_field] if by_grid else k
for f in glob.glob(os.path.join(out, tag + "_" + what + "_trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[key(r, "Grid_Size_X")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
fetch = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, tag + "_" + what + "_fetch", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            fetch[key(r, "Grid_Size")].append(float(r["Counter_Value"]))
path = os.path.join(out, tag + ("_select_kernels.csv" if what == "sel" else "_" + what + "_kernels.csv"))
with open(path, "w") as fo:
    fo.write("# tools/rocprof_%s.sh: kernel-trace durations (ns) and, from a separate --pmc pass, FETCH_SIZE per launch\n" % ("select" if what == "sel" else what))
    fo.write("# FETCH_SIZE in KB as reported; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction for wide streaming reads)\n")
    fo.write("kernel,calls,avg_ns,min_ns,max_ns,fetch_size_KB_avg,fetch_size_KB_max,read_bytes_max\n")
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        d = dur[k]
        fr = fetch.get(k, [])
        fo.write('"%s",%d,%.1f,%d,%d,%s,%s,%s\n' % (k, len(d), sum(d) / len(d), min(d), max(d),
                 "%.1f" % (sum(fr) / len(fr)) if fr else "", "%.1f" % max(fr) if fr else "",
                 "%.0f" % (max(fr) * 2048) if fr else ""))
print(open(path).read())
