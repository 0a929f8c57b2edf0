#!/bin/bash
# All profiling passes of a round on the GPU box (each rocprofv3 run is its own process, counters in runs of
# their own as MI355X_MICROARCH.md's rocprofv3 section prescribes): tools/prof_all.sh <out dir> [legs...]
set -u
OUT=${1:-gpurun_out/prof}; shift
LEGS=${@:-fht dhtgen inflate_zlib6 inflate_wg inflate_own inflate_stream c5}
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for leg in $LEGS; do
  for pass in stats fetch write sq1 sq2; do
    case $pass in
      stats) args="--kernel-trace --stats";;
      fetch) args="--pmc FETCH_SIZE";;
      write) args="--pmc WRITE_SIZE";;
      sq1)   args="--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS";;
      sq2)   args="--pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR";;
    esac
    # SQ counters: the deflate legs and the batched inflate only
    if [[ $pass == sq* && $leg == inflate_stream ]]; then continue; fi
    d=$OUT/${leg}_$pass
    rm -rf $d
    rocprofv3 $args --output-format csv -d $d -- python3 tools/prof_workload.py $leg > $OUT/${leg}_$pass.log 2>&1
    tail -1 $OUT/${leg}_$pass.log | grep -q '"leg"' || { echo "FAILED: $leg $pass"; tail -3 $OUT/${leg}_$pass.log; }
  done
done
echo done
