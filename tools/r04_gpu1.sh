#!/bin/bash
# round 4, GPU call 1: lab measurement of the LDS full-histogram idea, the new tests, the whole suite, the bench
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r04a}
( timeout 120 tools/lab/hist16_lab ) > gpurun_out/${T}_hist16_lab.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_r04.py tests/test_gpu_dist2.py tests/test_gpu_bench_n2.py -q 2>&1 | tail -40 ) > gpurun_out/${T}_newtests.log 2>&1
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_r04.py --deselect tests/test_gpu_dist2.py --deselect tests/test_gpu_bench_n2.py 2>&1 | tail -15 ) > gpurun_out/${T}_alltests.log 2>&1
( timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err )
cat gpurun_out/${T}_hist16_lab.log
tail -25 gpurun_out/${T}_newtests.log
tail -5 gpurun_out/${T}_alltests.log
tail -3 gpurun_out/${T}_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
    ex = d["extras"]
    print("value", d["value"], "ms_per_step", d["ms_per_step"])
    print("roofline", {k: v for k, v in d["roofline"].items() if k.startswith("frac") or k.startswith("kernel_avg") or k == "clocks_during_windows"})
    print("gates", ex.get("all_config_gates_pass"))
    for k, v in ex["configs"]["config1_resnet18_minmax_trt"].items():
        print("config1", k, v["us"], v["frac"], v["parity"], v.get("through_quantizer_forward_us"))
except Exception as e:
    print("bench parse failed", e)
PY
