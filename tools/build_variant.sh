#!/bin/bash
# Builds power-gzip_amd/libnxz_engine_<tag>.so from the working tree with extra compiler flags for ONE kernel file
# (an A/B candidate for tools/ab_lz77.py).  usage: tools/build_variant.sh <tag> <file.hip> [flags ...]
set -e
cd "$(dirname "$(readlink -f "$0")")/../power-gzip_amd/csrc"
tag=$1; f=$2; shift 2
mkdir -p build_base
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I . "$@" -c $f -o build_base/$f.$tag.o
objs=""
for o in build/*.o; do b=$(basename $o); [ "$b" = "$f.o" ] && objs="$objs build_base/$f.$tag.o" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnxz_engine_$tag.so $objs -lpthread
echo built libnxz_engine_$tag.so
