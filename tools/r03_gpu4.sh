#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_r03.py -x -q -k "group" 2>&1 | tail -25 ) > gpurun_out/r03d_tests.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err
tail -25 gpurun_out/r03d_tests.log; tail -c 3000 gpurun_out/r03d_bench.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03d_bench.json").read().strip().splitlines()[-1])
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "roofline", "parity_checked")}, indent=0)[:1500])
    print(json.dumps(d["extras"].get("configs"), indent=0)[:6000])
    print(json.dumps(d["extras"].get("model_wide_calibration"), indent=0)[:3000])
    print(json.dumps(d["cpu_baseline"], indent=0)[:1500])
except Exception as e:
    print("no bench line:", e)
PY
