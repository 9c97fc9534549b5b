#!/bin/bash
# differential soak of the new LZ4 decoder: index path forced on for every batch size
export CRYO_LZ4_INDEX_MIN=0
timeout 400 python3 tests/stress_gpu.py 150 11 2>&1 | tail -4
timeout 400 python3 tests/stress_gpu.py fuzz 150 12 2>&1 | tail -4
unset CRYO_LZ4_INDEX_MIN
timeout 300 python3 tests/stress_gpu.py 90 13 2>&1 | tail -3
timeout 300 python3 tests/stress_gpu.py fuzz 90 14 2>&1 | tail -3
