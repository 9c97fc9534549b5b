#!/bin/bash
CRYO_LZ4_STATS=1 python3 bench.py --workload lz4_decode --no-cpu-baseline --steps 1 --warmup 0 --blocks 32768 2>&1 | grep "lz4 seq" | cut -c1-400
