#!/bin/bash
python3 bench.py --workload lz4_decode --no-cpu-baseline --steps 1 --warmup 0 --blocks 65536 2>&1 | grep "idx " | head -4
python3 bench.py --workload lz4_decode --no-cpu-baseline --steps 1 --warmup 0 --blocks 16384 2>&1 | grep "idx " | head -4
