#!/bin/bash
bash profiles/collect.sh r02 lz4_decode > gpurun_out/collect_lz4_decode.log 2>&1
bash profiles/scripts/pmc_sq.sh r02_lz4_dec lz4_decode > gpurun_out/collect_sq.log 2>&1
tail -n 3 gpurun_out/collect_lz4_decode.log
