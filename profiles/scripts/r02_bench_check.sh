#!/bin/bash
# functional check of every bench.py mode (small sizes)
set -x
python3 bench.py --steps 3 --warmup 1 --blocks 16384 --cpu-blocks 512 2>&1 | tail -1 | cut -c1-1800
python3 bench.py --workload zstd_decode --steps 2 --warmup 1 --blocks 8192 --cpu-blocks 256 2>&1 | tail -1 | cut -c1-1500
python3 bench.py --workload lz4 --steps 2 --warmup 1 --blocks 8192 --cpu-blocks 256 2>&1 | tail -1 | cut -c1-2200
python3 bench.py --workload zstd --steps 2 --warmup 1 --blocks 8192 --cpu-blocks 256 2>&1 | tail -1 | cut -c1-2200
python3 bench.py --workload mixed --steps 3 --warmup 1 --blocks 4096 2>&1 | tail -1 | cut -c1-1500
python3 bench.py --gpus 2 --steps 3 --warmup 1 --blocks 16384 2>&1 | tail -1 | cut -c1-1200
