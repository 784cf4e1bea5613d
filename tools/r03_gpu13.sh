#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r03p}
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/${T}_tests.log 2>&1
( timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err )
tail -4 gpurun_out/${T}_tests.log; python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"])
ex = d.get("extras", {})
print("gates", ex.get("all_config_gates_pass"))
for k, v in ex.get("configs", {}).items(): print(k, json.dumps(v)[:600])
print(json.dumps(ex.get("model_wide_calibration"))[:800])
PY
