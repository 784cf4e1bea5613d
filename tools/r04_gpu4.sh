#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r04j}
( timeout 900 python -m pytest tests/test_gpu_r04.py tests/test_gpu_dist2.py tests/test_gpu_bench_n2.py -q 2>&1 | tail -30 ) > gpurun_out/${T}_newtests.log 2>&1
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_r04.py --deselect tests/test_gpu_dist2.py --deselect tests/test_gpu_bench_n2.py 2>&1 | tail -8 ) > gpurun_out/${T}_alltests.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err )
tail -25 gpurun_out/${T}_newtests.log
tail -4 gpurun_out/${T}_alltests.log
tail -3 gpurun_out/${T}_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
    ex = d["extras"]
    print("value", d["value"], "ms_per_step", d["ms_per_step"])
    print("roofline", {k: v for k, v in d["roofline"].items() if k.startswith("frac") or k.startswith("kernel_avg_us")})
    print("gates", ex.get("all_config_gates_pass"))
    def walk(dd, pre=""):
        for k, v in dd.items():
            if isinstance(v, dict):
                if "us" in v and "frac" in v:
                    print(pre + k, v["us"], v["frac"], v.get("parity"))
                walk(v, pre + k + "/")
    walk(ex["configs"]); walk(ex["model_wide_calibration"])
except Exception as e:
    print("bench parse failed", e)
PY
