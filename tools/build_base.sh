#!/bin/bash
# Builds power-gzip_amd/libnxz_engine_<tag>.so with ONE kernel file taken from a git revision (default HEAD), the rest from the
# working tree's objects: the baseline of an A/B run (tools/ab_lz77.py).  usage: tools/build_base.sh [file.hip] [rev] [tag]
set -e
cd "$(dirname "$0")/../power-gzip_amd/csrc"
f=${1:-nxz_lz77.hip}; rev=${2:-HEAD}; tag=${3:-base}
mkdir -p build_base
git show "$rev:power-gzip_amd/csrc/$f" > build_base/$f
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I . -c build_base/$f -o build_base/$f.o
objs=""
for o in build/*.o; do b=$(basename $o); [ "$b" = "$f.o" ] && objs="$objs build_base/$f.o" || objs="$objs $o"; done
objs=$(echo $objs | tr ' ' '\n' | grep -v nxz_deflate.hip.o | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnxz_engine_$tag.so $objs -lpthread
echo built ../libnxz_engine_$tag.so
