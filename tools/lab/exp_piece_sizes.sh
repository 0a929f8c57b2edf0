#!/bin/bash
# inflate() in steps and one-shot nx_uncompress calls against the size of the pieces a part of a stream is cut into
# (NXZ_PINFLATE_PIECE_BITS: the shortest piece, bits of the stream; NXZ_PINFLATE_PIECES_SHORT: how many pieces a short part is cut into at least).
#   usage: exp_piece_sizes.sh <out file> <file.gz>
out=${1:-gpurun_out/piece_sizes.txt}; gz=$2
f=tests/golden/alice29.txt
: > "$out"
run() {
	local name=$1; shift
	env "$@" timeout 120 power-gzip_amd/inflate_steps $gz 64 256 1024 2>&1 | grep "steps of" | sed "s/^/$name /; s/zlib, this.*//" >> "$out"
	for T in 1 16; do
		for kib in 256 1024; do
			per=$(( (8 << 20) / (kib * T) )); [ $per -gt 256 ] && per=256
			line=$(env "$@" timeout 120 power-gzip_amd/compdecomp_th $f $T $kib $per 2>&1 | grep '^{' | tail -1 | sed 's/.*"decompress_GiB_s": \([0-9.]*\), "decompress_us_per_call": \([0-9.]*\).*/decomp \1 GiB\/s \2 us a call/')
			echo "$name  ${kib} KiB x $T threads: $line" >> "$out"
		done
	done
}
run "default (4096 bits, 768)" NXZ_NOP=1
run "piece bits 2048         " NXZ_PINFLATE_PIECE_BITS=2048
run "piece bits 8192         " NXZ_PINFLATE_PIECE_BITS=8192
run "piece bits 16384        " NXZ_PINFLATE_PIECE_BITS=16384
run "2048 bits, 1536 pieces  " NXZ_PINFLATE_PIECE_BITS=2048 NXZ_PINFLATE_PIECES_SHORT=1536
run "4096 bits, 384 pieces   " NXZ_PINFLATE_PIECES_SHORT=384
run "4096 bits, 1536 pieces  " NXZ_PINFLATE_PIECES_SHORT=1536
cat "$out"
