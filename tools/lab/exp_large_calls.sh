#!/bin/bash
# Sixteen and 64 threads of nx_compress2 / nx_uncompress calls of 8 to 64 MiB (tools/compdecomp_th.c), calls beyond the merge's
# limit cut into slices (the default) and on lanes of their own (NXZ_MERGE_SLICES=0).   usage: exp_large_calls.sh <out file>
out=${1:-gpurun_out/large_calls.txt}
f=tests/golden/alice29.txt
: > "$out"
for sl in ${SLICES:-1 0}; do
	for T in 16 64; do
		for kib in ${SIZES:-8192 16384 32768 65536}; do
			per=$(( (8 << 20) / (kib * T) )); [ $per -lt 4 ] && per=4
			[ $(( kib * T * per )) -gt $(( 24 << 20 )) ] && continue
			line=$(NXZ_MERGE_SLICES=$sl timeout 200 power-gzip_amd/compdecomp_th $f $T $kib $per 2>&1 | grep '^{' | tail -1 | sed 's/.*"compress_GiB_s": \([0-9.]*\),.*"decompress_GiB_s": \([0-9.]*\),.*"bad": \([0-9]*\).*/comp \1  decomp \2 GiB\/s  bad \3/')
			echo "slices=$sl  ${kib} KiB x $T threads x $per: $line" >> "$out"
		done
	done
done
cat "$out"
