#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_r03.py tests/test_gpu_r02.py tests/test_gpu_parity.py -x -q -k "group or mse or backward or lsq or calib or r02 or fuzz" 2>&1 | tail -8 ) > gpurun_out/r03i_tests.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r03i_bench.json 2> gpurun_out/r03i_bench.err
tail -4 gpurun_out/r03i_tests.log; tail -c 800 gpurun_out/r03i_bench.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03i_bench.json").read().strip().splitlines()[-1])
    e = d["extras"]
    print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_avg_us_1024_window_last"], e["all_config_gates_pass"])
    print({k: v for k, v in e.items() if not isinstance(v, dict)})
    def walk(p, x):
        if isinstance(x, dict):
            if "us" in x and "parity" in x:
                print(p, x["us"], x["frac"], x["parity"], {k: v for k, v in x.items() if k in ("one_by_one_us", "one_launch_per_matrix_us", "us_per_4096x4096_equivalent", "end_to_end_us", "frac_of_fp32_vector_peak", "valu_tflops")})
            else:
                for k, v in x.items(): walk(p + "/" + k, v)
    walk("configs", e.get("configs")); walk("mw", e.get("model_wide_calibration"))
except Exception as ex:
    print("no bench line:", ex)
PY
