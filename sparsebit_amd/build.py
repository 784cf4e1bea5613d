"""Ahead-of-time build of libsbq.so (HIP, gfx950 only) -- no import-time JIT.

The reference JIT-compiles its extension when the package is imported
(sparsebit/quantization/quantizers/quant_tensor.py:7-22); here the shared
library is built once, in-tree, by `python -m sparsebit_amd.build` (or
`__graft_entry__.build()`), and `sparsebit_amd.lib` only ever dlopens it.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libsbq.so")
ARCH = "gfx950"

# -ffp-contract=off: the parity contract is op-for-op IEEE fp32 (no fused
# multiply-add may be formed across the reference's separate torch ops).
FLAGS = [
    "--offload-arch=" + ARCH,
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fPIC",
    "-fvisibility=hidden",
    "-Wall",
    "-Wno-unused-function",
    # the first 16 dwords of a kernel's (scalar) arguments arrive in SGPRs with the wave instead of through a scalar
    # load from the argument block: ~0.3 us off every launch's critical path (tools/lab/kernarg_lab.hip; the headline
    # QDQ kernel 11.05 -> 10.76 us)
    "-mllvm",
    "-amdgpu-kernarg-preload-count=16",
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libsbq.so cannot be built")
    return exe


# per-file extras.  sbq_qdq.hip: preload the first 14 kernarg dwords into SGPRs (see the comment
# on qdq_pack_kernel)
EXTRA_FLAGS = {
    "sbq_qdq.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=14"],
    "sbq_qdq_resident.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=14"],
    # calib_mse_lanes_kernel: left alone, the SLP vectoriser pairs the loop's fmas into v_pk_fma_f32 and pays six
    # v_mov per eight evaluations to line the operands up -- packed fp32 issues at 6-8 cycles against 4 for two plain
    # ones (tools/lab/valu_rate.hip), so the pairing is a loss here
    "sbq_calib.hip": ["-fno-slp-vectorize"],
}
# files compiled more than once: (object suffix, extra flags) per additional unit.  sbq_select_win.hip instantiates
# the one-launch selection engine per input type in a unit of its own (SBQ_WIN_PART, see the top of that file): the
# three units compile in parallel instead of one after the other (232 s -> ~80 s on the critical path of a clean build)
EXTRA_UNITS = {
    "sbq_select_win.hip": [("_bf16", ["-DSBQ_WIN_PART=1"]), ("_f16", ["-DSBQ_WIN_PART=2"])],
}


# development only: extra hipcc flags for every file (e.g. SBQ_EXTRA_HIPCC_FLAGS="-DSBQ_SEL_STAMPS=1")
FLAGS += os.environ.get("SBQ_EXTRA_HIPCC_FLAGS", "").split()


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "sbq.h"))
    headers.append(os.path.abspath(__file__))
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj])
        for suffix, flags in EXTRA_UNITS.get(src, []):
            obj2 = os.path.join(OBJ, src[:-4] + suffix + ".o")
            objs.append(obj2)
            if force or _stale(obj2, [os.path.join(CSRC, src)] + headers):
                jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + flags + ["-c", os.path.join(CSRC, src), "-o", obj2])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout)
        if verbose and r.stdout.strip():
            print(r.stdout)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
