#!/bin/bash
bash profiles/collect.sh r02 lz4_decode > gpurun_out/collect_lz4_decode.log 2>&1
bash profiles/scripts/pmc_sq.sh r02_lz4_dec lz4_decode > gpurun_out/collect_sq.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_driver_like.json 2> gpurun_out/r02_driver_like.err
python3 bench.py --workload mixed > gpurun_out/r02_mixed.json 2> gpurun_out/r02_mixed.err
for d in narrow int4 random zeros; do python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 --dist $d 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$d', j['value'], j['roofline']['frac'], j['config']['compression_ratio'])"; done > gpurun_out/r02_dists.txt 2>&1
python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --blocks 8192 --block-size 1048576 2>&1 | tail -1 | cut -c1-700 > gpurun_out/r02_1mib.txt
tail -c 1500 gpurun_out/r02_driver_like.json; cat gpurun_out/r02_dists.txt gpurun_out/r02_1mib.txt; tail -c 1200 gpurun_out/r02_mixed.json
