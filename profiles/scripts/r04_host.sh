#!/bin/bash
# round 4: host-buffer API rates, several calls each (median, min-max); topology; restricted to the GPU's local CPUs; through the
# multi-GPU dispatcher with 1 and 2 handles
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_host
{
timeout 900 python3 -m pytest tests/test_gpu_host.py tests/test_gpu_multi.py tests/test_gpu_pool.py -x -q 2>&1 | tail -3
echo "== all cpus the process may use"
timeout 900 python3 profiles/host_api_rate.py
LOCAL=$(python3 - <<'PY'
import glob
for d in sorted(glob.glob("/sys/bus/pci/devices/*")):
    try:
        if open(d + "/vendor").read().strip() == "0x1002" and open(d + "/class").read().startswith("0x12"):
            print(open(d + "/local_cpulist").read().strip()); break
    except OSError: pass
PY
)
echo "== restricted to the GPU's local cpus ($LOCAL)"
[ -n "$LOCAL" ] && HOST_API_CPUS=$LOCAL HOST_API_REPS=5 timeout 900 python3 profiles/host_api_rate.py 2>&1 | grep -v "ONE block"
echo "== multi-GPU dispatcher"
HOST_API_MULTI=1 HOST_API_REPS=5 timeout 900 python3 profiles/host_api_rate.py
} 2>&1 | tee gpurun_out/r04_host/out.txt
