#!/bin/bash
# Sixteen threads of nx_uncompress calls of 256 KiB and 1 MiB (tools/compdecomp_th.c) under the settings that could bear on why
# the callers' small kernels do not run side by side.   usage: exp_inflate_threads.sh <out file>
out=${1:-gpurun_out/inflate_threads.txt}
f=tests/golden/alice29.txt
: > "$out"
run() {
	local name=$1; shift
	for kib in 256 1024; do
		per=$(( (8 << 20) / (kib * 16) )); [ $per -gt 512 ] && per=512
		line=$(env "$@" timeout 120 power-gzip_amd/compdecomp_th $f 16 $kib $per 2>&1 | grep '^{' | tail -1 | sed 's/.*"decompress_GiB_s": \([0-9.]*\), "decompress_us_per_call": \([0-9.]*\).*/\1 GiB\/s \2 us a call/')
		echo "$name  ${kib} KiB x 16 threads: $line" >> "$out"
	done
}
run "default            " NXZ_NOP=1
run "zero copy off      " NXZ_PINFLATE_ZEROCOPY=0
run "gate 4             " NXZ_PARALLEL_INFLATE_MAX=4
run "gate 8             " NXZ_PARALLEL_INFLATE_MAX=8
run "gate none          " NXZ_PARALLEL_INFLATE_MAX=0
run "no own stream      " NXZ_PINFLATE_OWN_STREAM=0
run "no host stage      " NXZ_HOST_STAGE=0
run "serialize kernels  " AMD_SERIALIZE_KERNEL=3
run "no direct dispatch " AMD_DIRECT_DISPATCH=0
run "queues 2           " GPU_MAX_HW_QUEUES=2
run "queues 1           " GPU_MAX_HW_QUEUES=1
run "job path (min 16M) " NXZ_PARALLEL_INFLATE_MIN=16777216
cat "$out"
