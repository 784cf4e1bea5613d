"""Dev tool: build tools/lab/libsbq_variant.so -- the product library with extra -D flags on the selection engine's three
units (e.g. -DSBQ_POLL_SLEEP=2), everything else from the regular object files.  tools/lab/h16_time.py loads it when
SBQ_LIB=<path> is set."""
import concurrent.futures, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import build as B
B.build()
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(B.CSRC, "sbq_select_win.hip")
units = [("", [])] + B.EXTRA_UNITS["sbq_select_win.hip"]
def compile_unit(u):
    suffix, flags = u
    obj = "/tmp/sel_variant%s.o" % suffix
    subprocess.check_call([B._hipcc()] + B.FLAGS + flags + sys.argv[1:] + ["-c", src, "-o", obj])
    return obj
with concurrent.futures.ThreadPoolExecutor(max_workers=3) as ex:
    objs = list(ex.map(compile_unit, units))
regular = [os.path.join(B.OBJ, f[:-4] + ".o") for f in B.sources() if f != "sbq_select_win.hip"]
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(here, "libsbq_variant.so")] + regular + objs)
print("built", os.path.join(here, "libsbq_variant.so"))
