#!/bin/bash
# The threaded harness (tools/compdecomp_th.c) on the engine alone, with the HIP runtime's hardware queue count as given
# and as the default leaves it: threads x buffer sizes, one line a run.   usage: exp_hw_queues.sh <out file> [queue counts ...]
out=${1:-gpurun_out/hw_queues.txt}; shift
qs=${@:-"0 8"}
f=tests/golden/alice29.txt
: > "$out"
for q in $qs; do
	for T in 16 64; do
		for kib in 64 256 1024 4096 16384; do
			per=$(( (8 << 20) / (kib * T) )); [ $per -gt 512 ] && per=512; [ $per -lt 8 ] && per=8
			if [ "$q" = 0 ]; then line=$(timeout 120 power-gzip_amd/compdecomp_th $f $T $kib $per 2>&1 | grep '^{' | tail -1)
			else line=$(GPU_MAX_HW_QUEUES=$q timeout 120 power-gzip_amd/compdecomp_th $f $T $kib $per 2>&1 | grep '^{' | tail -1); fi
			echo "queues=$q $line" >> "$out"
		done
	done
done
cat "$out"
