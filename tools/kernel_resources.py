"""Register / LDS / scratch budget of every kernel in sparsebit_amd/libsbq.so, read from the code objects' metadata
(no GPU needed):  python tools/kernel_resources.py [--spills]
A kernel that spills vector registers to scratch reloads them through the same in-order vector-memory path as its
data loads -- in round 6 that was 10 us of the 62 us model-wide selection (group_kth_kernel).
tests/test_kernel_resources.py keeps the hot kernels at zero spills."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def kernels(lib=None):
    """[{name (demangled), vgpr_count, ..., group_segment_fixed_size}] for every kernel of the library."""
    lib = lib or os.path.join(ROOT, "sparsebit_amd", "libsbq.so")
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)  # (llvm-objdump writes the bundles next to its input)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True,
                                   capture_output=True, text=True).stdout
            cur = None
            for line in notes.splitlines():
                if line.startswith("  - ."):  # a new kernel record (its argument records are indented further)
                    cur = {}
                    out.append(cur)
                m = re.match(r"  (?:- |  )\.(\w+):\s+(.*)$", line)
                if not m or cur is None:
                    continue
                key, val = m.group(1), m.group(2).strip().strip("'")
                if key == "name":
                    cur["mangled"] = val
                elif key in FIELDS:
                    cur[key] = int(val)
    out = [k for k in out if "mangled" in k]
    names = subprocess.run([shutil.which("c++filt") or "c++filt"], input="\n".join(k["mangled"] for k in out), capture_output=True,
                           text=True, check=True).stdout.splitlines()
    for k, nm in zip(out, names):
        nm = nm.replace("sbq::(anonymous namespace)::", "").replace("sbq::", "")
        k["name"] = re.sub(r"^void ", "", nm.split("(")[0])
    return out


if __name__ == "__main__":
    only_spills = "--spills" in sys.argv
    rows = kernels()
    print("%-78s %5s %5s %7s %7s %8s %7s" % ("kernel", "vgpr", "sgpr", "vspill", "sspill", "scratch", "LDS"))
    for k in sorted(rows, key=lambda r: r["name"]):
        if only_spills and not k.get("vgpr_spill_count") and not k.get("private_segment_fixed_size"):
            continue
        print("%-78s %5d %5d %7d %7d %8d %7d" % (k["name"][:78], k.get("vgpr_count", 0), k.get("sgpr_count", 0),
                                                 k.get("vgpr_spill_count", 0), k.get("sgpr_spill_count", 0),
                                                 k.get("private_segment_fixed_size", 0), k.get("group_segment_fixed_size", 0)))
    print("%d kernels" % len(rows))
