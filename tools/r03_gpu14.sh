#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r03x}
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r02.py tests/test_gpu_r03.py -x -q -k "mse or calibration or group" 2>&1 | tail -4 ) > gpurun_out/${T}_tests.log 2>&1
( timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err )
tail -3 gpurun_out/${T}_tests.log; python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
ex = d["extras"]
print("gates", ex.get("all_config_gates_pass"), d["value"], d["roofline"]["frac"])
c2 = ex["configs"]["config2_mse_per_channel"]; print("config2", c2["us"], c2["end_to_end_us"], c2["frac_of_fp32_vector_peak"], c2["parity"])
c3 = ex["configs"]["config3_percentile"]; print("config3", {k: v["us"] for k, v in c3.items() if isinstance(v, dict)})
print("model-wide mse", ex["model_wide_calibration"]["mse_qparams"]["us"], ex["model_wide_calibration"]["mse_qparams"]["parity"])
PY
