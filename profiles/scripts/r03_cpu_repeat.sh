#!/bin/bash
# the driver's bench invocation three times in a row: value, frac, cpu_baseline on 1 thread and on every physical core
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for i in 1 2 3 4; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03_bench_driver_args_$i.json; done
python3 - <<'PY'
import json
for i in (1, 2, 3, 4):
    d = json.load(open("gpurun_out/r03_bench_driver_args_%d.json" % i))
    c = d["cpu_baseline"]
    print("run %d: value %.1f GB/s, %.3f ms/step, frac %.4f | cpu 1 thread %.2f, %d pinned physical cores %.1f GB/s" %
          (i, d["value"], d["ms_per_step"], d["roofline"]["frac"], c["value"], c.get("cores_used", 0), c["all_cores_value"]))
PY
