#!/bin/bash
for a in 0 1 2 4 7; do echo "== ablate $a"; CRYO_LZ4_ABLATE=$a CRYO_LZ4_STATS=1 python3 bench.py --no-cpu-baseline --no-verify --steps 1 --warmup 0 --blocks 32768 2>&1 | grep -E "seq cycles|seq stats|avg_launch" | sed 's/.*avg_launch_ms": \([0-9.]*\).*/avg_launch_ms \1/' ; done
