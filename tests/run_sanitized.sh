#!/bin/bash
# CPU suite with AddressSanitizer + UBSan builds of the oracle and the host library (GPU ASan is not
# available on this pool; the kernels are covered by the differential stress runs instead).
# Works on a scratch copy: bash tests/run_sanitized.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cp -r "$ROOT"/pg_cryogen_amd "$ROOT"/oracle "$ROOT"/include "$ROOT"/tests "$ROOT"/pg "$ROOT"/tools "$T"/
cd "$T"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer"
make -s -C oracle clean
make -s -C oracle CFLAGS="-O1 -g -fPIC -Wall -std=c99 $SAN"
gcc -shared -o oracle/libcryo_oracle.so oracle/*.o -lpthread -ldl -lm $SAN
make -s -C pg_cryogen_amd/host clean
make -s -C pg_cryogen_amd/host CFLAGS="-O1 -g -fPIC -Wall -std=gnu11 -I../../include $SAN"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
# the encoder oracle against the live libzstd on random blocks, every level, with the sanitizers watching
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
python tests/hunt_oracle.py ${HUNT_SECONDS:-60} 1
rm -rf "$T"
