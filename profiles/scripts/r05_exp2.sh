#!/bin/bash
# round 5, zstd level-1 encoder experiments: where the `fast` table lives
#   prod                      u32 table in the workgroup's global workspace, 4 096 waves (16 per CU)
#   zdbg GRID=n               the same with n waves: 768 x 32 KiB = 24 MiB of tables (what the L2s hold together)
#   zdbg CRYO_ZSTD_ENC_LDS=1  table in LDS (32 KiB of u32 entries + the entropy stage's 9 KiB: 3 waves per CU = 768 waves)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r05_zstd_enc_lds.txt; : > $out
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 600 python3 bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/err.txt | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); c = d['config']
    print('%-44s encode %6.2f GB/s  decode %6.1f GB/s  ratio %s' % ('$label', c['encode_GBps'], c['decode_GBps'], c['compression_ratio']))
except Exception as e:
    print('%-44s FAILED %s' % ('$label', e)); print(open('gpurun_out/err.txt').read()[-600:])" >> $out
}
V=$PWD/profiles/variants_zdbg.so
run "prod (global table, 4096 waves)" A=1
run "debug build, same" CRYO_CODEC_LIB=$V
run "global table, 3072 waves" CRYO_CODEC_LIB=$V CRYO_ZSTD_ENC_GRID=3072
run "global table, 2048 waves" CRYO_CODEC_LIB=$V CRYO_ZSTD_ENC_GRID=2048
run "global table, 1536 waves" CRYO_CODEC_LIB=$V CRYO_ZSTD_ENC_GRID=1536
run "global table, 768 waves (24 MiB of tables)" CRYO_CODEC_LIB=$V CRYO_ZSTD_ENC_GRID=768
run "global table, 512 waves (16 MiB of tables)" CRYO_CODEC_LIB=$V CRYO_ZSTD_ENC_GRID=512
run "LDS table (u32), 768 waves" CRYO_CODEC_LIB=$V CRYO_ZSTD_ENC_LDS=1
cat $out
# traffic of the two 768-wave forms
for v in "CRYO_ZSTD_ENC_GRID=768" "CRYO_ZSTD_ENC_LDS=1"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/r05_zenc_${v%%=*}_$c; rm -rf $d
    (cd /tmp && env TMPDIR=/tmp CRYO_CODEC_LIB=$V $v rocprofv3 --pmc $c --output-format csv -d $d -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload zstd --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1)
    python3 - <<PY >> $out
import csv, glob
f = glob.glob("$d/**/*counter_collection.csv", recursive=True)
tot = {}
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1][:24]
    tot.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in tot.items():
    if "zstd_enc" in k or "k_compare" in k: print("$v $c %-24s launches %d  %.2f GB per launch (raw KiB counter x 1024)" % (k, len(v), sum(v) / len(v) * 1024 / 1e9))
PY
  done
done
tail -12 $out
# the suite on this tree (nt streaming stores, new full-size bench tests, trim test)
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r05_suite1.txt
