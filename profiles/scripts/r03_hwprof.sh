#!/bin/bash
# where a k_zhufw wave and a k_zplan wave spend their time (profiling build of zstd_pipe.hip: -DCRYO_HW_PROF, s_memtime stamps
# summed into spare counters; printed by CRYO_ZSTD_STATS=1), and why blocks are handed back to k_zhuf.
# Build the variant first (on the build host):
#   cd pg_cryogen_amd/csrc && hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -DCRYO_HW_PROF -c zstd_pipe.hip -o /tmp/zp.o &&
#   g++ -shared -o ../../profiles/variants_hwprof.so $(ls *.o | grep -v zstd_pipe.o) /tmp/zp.o -Wl,--no-as-needed -lstdc++ -lm
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_hwprof
export CRYO_CODEC_LIB=$(pwd)/profiles/variants_hwprof.so CRYO_ZSTD_STATS=1 CRYO_ZSTD_LANES=1
timeout 300 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 1 --warmup 0 "$@" 2>&1 | grep "zstd pipe" | cut -c1-300 | head -8 | tee gpurun_out/r03_hwprof/log.txt
