#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r04k}
( timeout 900 python -m pytest tests/test_gpu_r04.py tests/test_gpu_bench_n2.py -q -x 2>&1 | tail -12 ) > gpurun_out/${T}_newtests.log 2>&1
( timeout 200 python tools/lab/r04_probe.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${T}_probe.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err )
tail -12 gpurun_out/${T}_newtests.log
cat gpurun_out/${T}_probe.log
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
    ex = d["extras"]
    print("value", d["value"], "frac", d["roofline"]["frac"], "gates", ex.get("all_config_gates_pass"))
    c1 = ex["configs"]["config1_resnet18_minmax_trt"]["minmax_observer_4_batches_64x64x56x56_fp32"]
    print("config1 minmax", c1["us"], c1["frac"], c1["parity"], c1["sbq_channel_stats_two_launches_per_batch_us"])
    c2 = ex["configs"]["config2_mse_per_channel"]["per_tensor_histogram_route"]
    print("mse16", c2["us"], c2["parity"], c2["per_element_route_us"])
except Exception as e:
    print("bench parse failed", e)
PY
