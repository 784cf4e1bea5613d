#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_r03.py tests/test_gpu_observe_fused.py tests/test_gpu_gptq_stress.py tests/test_gpu_r02.py -x -q 2>&1 | tail -15 ) > gpurun_out/r03f_tests.log 2>&1
( timeout 600 python tools/r03_gptq_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 ) > gpurun_out/r03f_gptq.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err
tail -8 gpurun_out/r03f_tests.log; cat gpurun_out/r03f_gptq.log; tail -c 1500 gpurun_out/r03f_bench.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03f_bench.json").read().strip().splitlines()[-1])
    e = d["extras"]
    print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_avg_us_1024_window_last"])
    print({k: v for k, v in e.items() if not isinstance(v, dict)})
    def walk(p, x):
        if isinstance(x, dict):
            if "us" in x and "parity" in x:
                print(p, x["us"], x["frac"], x["parity"], {k: v for k, v in x.items() if k in ("one_by_one_us", "one_launch_per_matrix_us", "us_per_4096x4096_equivalent", "end_to_end_us", "cache_resident_us", "gs_max_rel_err")})
            else:
                for k, v in x.items(): walk(p + "/" + k, v)
    walk("configs", e.get("configs")); walk("mw", e.get("model_wide_calibration"))
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["matches_gpu_output"], d["cpu_baseline"]["other_legs"])
except Exception as ex:
    print("no bench line:", ex)
PY
