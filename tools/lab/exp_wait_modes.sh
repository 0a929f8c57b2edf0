#!/bin/bash
# How the host waits for the device (interrupt, active wait, polling) against what the calls through the zlib-style API take:
# one and sixteen threads of nx_compress2 / nx_uncompress (tools/compdecomp_th.c) and inflate() in steps (tools/inflate_steps.c).
#   usage: exp_wait_modes.sh <out file> <file.gz for the steps>
out=${1:-gpurun_out/wait_modes.txt}; gz=$2
f=tests/golden/alice29.txt
: > "$out"
run() {
	local name=$1; shift
	for T in 1 16; do
		for kib in 64 256 1024; do
			per=$(( (8 << 20) / (kib * T) )); [ $per -gt 256 ] && per=256
			line=$(env "$@" timeout 120 power-gzip_amd/compdecomp_th $f $T $kib $per 2>&1 | grep '^{' | tail -1 | sed 's/.*"compress_GiB_s": \([0-9.]*\),.*"decompress_GiB_s": \([0-9.]*\), "decompress_us_per_call": \([0-9.]*\).*/comp \1  decomp \2 GiB\/s \3 us a call/')
			echo "$name  ${kib} KiB x $T threads: $line" >> "$out"
		done
	done
	[ -n "$gz" ] && env "$@" timeout 120 power-gzip_amd/inflate_steps $gz 64 256 1024 2>&1 | grep "steps of" | sed "s/^/$name /; s/zlib, this.*//" >> "$out"
}
run "default        " NXZ_NOP=1
run "active wait 1ms" ROC_ACTIVE_WAIT_TIMEOUT=1000
run "no interrupts  " HSA_ENABLE_INTERRUPT=0
cat "$out"
