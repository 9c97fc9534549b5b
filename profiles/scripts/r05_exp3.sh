#!/bin/bash
# round 5: decode after encode per dispatch; blocks-per-call crossover against the host library; index walkers at mid-size
# batches; HBM traffic of the 131 072-block shape (the per-GPU share of configs[3]); traffic of the zstd LDS-table encoder
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
python3 profiles/scripts/r05_after_encode.py > gpurun_out/r05_decode_after_encode.txt 2>&1; cat gpurun_out/r05_decode_after_encode.txt
timeout 900 python3 profiles/crossover.py > gpurun_out/r05_crossover.txt 2>&1; tail -70 gpurun_out/r05_crossover.txt
out=gpurun_out/r05_walkers.txt; : > $out
for args in "--blocks 1024" "--blocks 1024 --lz4-walkers 64" "--blocks 1024 --lz4-walkers 16" "--blocks 2048" "--blocks 2048 --lz4-walkers 32" "--blocks 4096" "--blocks 4096 --lz4-walkers 32" "--blocks 256" "--blocks 256 --lz4-walkers 64"; do
  echo "== bench.py $args" >> $out
  python3 profiles/scripts/ab.py --prof --steps 30 --args "$args" prod >> $out 2>&1
done
cat $out
bash profiles/collect.sh r05n lz4_decode --blocks 131072 > gpurun_out/r05n_collect.log 2>&1; tail -3 gpurun_out/r05n_collect.log
V=$PWD/profiles/variants_zdbg.so
for v in "CRYO_ZSTD_ENC_GRID=768" "CRYO_ZSTD_ENC_LDS=1" "A=1"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$PWD/gpurun_out/r05_zenc_${v%%=*}_$c; rm -rf $d
    (cd /tmp && env TMPDIR=/tmp CRYO_CODEC_LIB=$V $v rocprofv3 --pmc $c --output-format csv -d $d -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload zstd --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1)
    python3 - <<PY
import csv, glob
f = glob.glob("$d/**/*counter_collection.csv", recursive=True)
tot = {}
for r in csv.DictReader(open(f[0])) if f else []:
    k = r["Kernel_Name"].split("(")[0].split("::")[-1][:24]
    tot.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in tot.items():
    if "zstd_enc" in k or "k_compare" in k: print("$v $c %-24s launches %d  %.2f GB per launch (raw KiB counter x 1024)" % (k, len(v), sum(v) / len(v) * 1024 / 1e9))
PY
  done
done > gpurun_out/r05_zenc_traffic.txt 2>&1
cat gpurun_out/r05_zenc_traffic.txt
