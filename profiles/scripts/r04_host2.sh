#!/bin/bash
# round 4: host-buffer API with the staging threads / pinned buffers on the GPU's NUMA node (CRYO_OPT_NUMA_LOCAL, default) and without
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_host2
{
timeout 900 python3 -m pytest tests/test_gpu_host.py tests/test_gpu_multi.py tests/test_gpu_pool.py -x -q 2>&1 | tail -3
python3 -c "
import torch; print('[topology] visible device pci bus id:', torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0),'pci_bus_id') else 'n/a')"
for i in 1 2 3; do
echo "== NUMA-local (default), run $i"
HOST_API_REPS=7 timeout 900 python3 profiles/host_api_rate.py 2>&1 | grep -v "ONE block\|\[topology\] 0000"
done
echo "== CRYO_OPT_NUMA_LOCAL = 0"
HOST_API_NUMA=0 HOST_API_REPS=7 timeout 900 python3 profiles/host_api_rate.py 2>&1 | grep -v "ONE block\|\[topology\] 0000"
echo "== multi-GPU dispatcher"
HOST_API_MULTI=1 HOST_API_REPS=5 timeout 900 python3 profiles/host_api_rate.py 2>&1 | grep -v "\[topology\] 0000"
} 2>&1 | tee gpurun_out/r04_host2/out.txt
