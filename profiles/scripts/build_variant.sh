#!/bin/bash
# A/B builds: profiles/scripts/build_variant.sh NAME "EXTRA FLAGS" file1.hip [file2.hip ...]
# recompiles only the named sources of pg_cryogen_amd/csrc with the extra flags and links them with the production objects
# into profiles/variants_NAME.so (loaded through CRYO_CODEC_LIB; never copied over the product library; *.so is git-ignored
# but travels to the GPU box).  Run `make -C pg_cryogen_amd/csrc` first.
set -e
NAME=$1; EXTRA=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd); C=$ROOT/pg_cryogen_amd/csrc; T=$ROOT/build/variants/$NAME; mkdir -p $T
OBJS=""
for o in $C/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do if [ "$(basename ${f%.*})" = "$b" ]; then use=$T/$b.o; fi; done
  OBJS="$OBJS $use"
done
for f in "$@"; do
  b=$(basename ${f%.*}); x=""; case $f in *.cpp) x="-x hip";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$ROOT/include -I$C -Wall -Wno-unused-function -Wno-pass-failed $EXTRA $x -c $C/$f -o $T/$b.o &
done
wait
g++ -shared -o $ROOT/profiles/variants_$NAME.so $OBJS -Wl,--no-as-needed -lstdc++ -lm
echo built profiles/variants_$NAME.so
