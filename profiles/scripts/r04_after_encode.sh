#!/bin/bash
# round 4: why is LZ4 decode slower right after an encode pass?  Per-step times of --workload lz4 with and
# without idle time between the passes, clocks and power sampled while it runs, next to the decode-only workload
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_after_encode
smi() { while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.25; done; }
{
echo "== decode only"
CRYO_BENCH_TRACE=1 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | grep -E "steps|value" | cut -c1-300
echo "== compress + decompress, back to back"
smi > gpurun_out/r04_after_encode/smi.txt & SMI=$!
CRYO_BENCH_TRACE=1 python3 bench.py --workload lz4 --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep -E "bench trace|value" | cut -c1-400
kill $SMI
echo "== compress + decompress, 300 ms idle between the passes"
CRYO_BENCH_GAP_MS=300 CRYO_BENCH_TRACE=1 python3 bench.py --workload lz4 --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep -E "bench trace|value" | cut -c1-400
echo "== compress + decompress, 20 ms idle"
CRYO_BENCH_GAP_MS=20 CRYO_BENCH_TRACE=1 python3 bench.py --workload lz4 --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep -E "bench trace|value" | cut -c1-400
echo "== clocks / power while the back-to-back run was going (every 0.25 s; first 40 samples with a busy GPU)"
grep -v "^$" gpurun_out/r04_after_encode/smi.txt | head -60
} 2>&1 | tee gpurun_out/r04_after_encode/out.txt
