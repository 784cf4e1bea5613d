"""Dev tool: build tools/lab/libsbq_variant.so -- the product library with extra -D flags on the selection engine's three
units (e.g. -DSBQ_POLL_SLEEP=2), everything else from the regular object files.  tools/lab/h16_time.py loads it when
SBQ_LIB=<path> is set.  `--file sbq_backward.hip -D...` applies the flags to that source file instead."""
import concurrent.futures, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import build as B
B.build()
here = os.path.dirname(os.path.abspath(__file__))
fname = "sbq_select_win.hip"
if "--file" in sys.argv:
    i = sys.argv.index("--file")
    fname = sys.argv[i + 1]
    del sys.argv[i:i + 2]
src = os.path.join(B.CSRC, fname)
units = [("", [])] + B.EXTRA_UNITS.get(fname, [])
def compile_unit(u):
    suffix, flags = u
    obj = "/tmp/sel_variant%s.o" % suffix
    subprocess.check_call([B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(fname, []) + flags + sys.argv[1:] + ["-c", src, "-o", obj])
    return obj
with concurrent.futures.ThreadPoolExecutor(max_workers=3) as ex:
    objs = list(ex.map(compile_unit, units))
regular = [os.path.join(B.OBJ, f[:-4] + ".o") for f in B.sources() if f != fname]
for suffix, _ in B.EXTRA_UNITS.get("sbq_select_win.hip", []) if fname != "sbq_select_win.hip" else []:
    regular.append(os.path.join(B.OBJ, "sbq_select_win" + suffix + ".o"))
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(here, "libsbq_variant.so")] + regular + objs)
print("built", os.path.join(here, "libsbq_variant.so"))
