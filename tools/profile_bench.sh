#!/bin/bash
# Profiles the bench command on the GPU box: one duration pass and separate counter passes (the microarchitecture
# guide: FETCH_SIZE and WRITE_SIZE cannot share a pass; counters only next to --kernel-trace), then reduces them ON
# THE BOX to profiles-style summaries (the raw per-dispatch CSVs are tens of MB per pass and are deleted).
#   bash tools/profile_bench.sh r03 "<bench args>" "<R,NW,chunk,pairs,WxH of the chunk kernel's schedule>" \
#        "<R,NW,k,pairs,WxH,warps,inner of the persistent kernel's schedule>"
#   -> gpurun_out/prof_r03/{counters.json, kernel_stats.csv, *.log}
set -e
TAG=${1:-r03}
ARGS=${2:---steps 3 --warmup 1 --no-cpu-baseline --whole-image-crop}
TILE=${3:-4,16,10,1,1024x436}
PTILE=${4:-3,16,5,1,1024x436,300,10}
OUT=gpurun_out/prof_$TAG
# One rank only: with --gpus N bench.py starts torch.distributed.run from a process the profiler's preloaded library has
# already initialised on the GPU (a launcher hop, and the counters would not be a rank's).  To profile a rank of a multi-GPU
# run put `python3 bench.py --gpus N ...` directly after `--` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by hand.
case " $ARGS " in *" --gpus "*) echo "profile_bench.sh profiles ONE rank: drop --gpus from the bench arguments" >&2; exit 2;; esac
export TMPDIR=/tmp
mkdir -p $OUT
run() { name=$1; shift; echo "== pass $name"; rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1; }
run stats --kernel-trace --stats
run sq1 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
run sq3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
python3 tools/pmc_summary.py --out $OUT/counters.json --label "$TAG: python3 bench.py $ARGS" --chunk-tile "$TILE" --persist-tile "$PTILE" \
    --grid tvl1_persist_kernel=253952 --grid tvl1_chunk_kernel=245760 --grid tvl1_warp_kernel=446464 --grid nl_fused_kernel=458752 \
    --grid tvl1_chunk_kernel@batch4=995328 --grid tvl1_warp_kernel@batch4=1785856 --tile tvl1_chunk_kernel@batch4=3,12,5,4,1024x436 \
    $OUT/stats $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/fetch $OUT/write
# (the bench starts child processes -- the drop-in executable of its end_to_end leg -- and each writes its own summary: keep the bench's,
# the largest one)
cp "$(find $OUT/stats -name "*kernel_stats.csv" -printf '%s %p\n' | sort -n | tail -1 | cut -d' ' -f2-)" $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/fetch $OUT/write
echo done
