#!/bin/bash
# Profiles the bench command on the GPU box: one duration pass and separate counter passes (the microarchitecture
# guide: FETCH_SIZE and WRITE_SIZE cannot share a pass; counters only next to --kernel-trace).  Usage (from the repo root):
#   bash tools/profile_bench.sh r02 "<bench args>"      -> gpurun_out/prof_r02/{stats,sq1,sq2,fetch,write,...}
set -e
TAG=${1:-r02}
ARGS=${2:---steps 2 --warmup 1 --no-cpu-baseline}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
run() { name=$1; shift; echo "== pass $name"; rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1; }
run stats --kernel-trace --stats
run sq1 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
run sq3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
echo done
