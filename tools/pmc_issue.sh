#!/bin/bash
# Counters behind bench.py's `issue_roofline` and `roofline.traffic`, per workload, in separate rocprofv3 --pmc passes (a pass =
# one run of bench.py with its cpu_baseline and `configs` legs off and no pre-roll). --kernel-trace rides along so that every
# profiled dispatch also has its own begin/end time: the clock the kernel ran at UNDER the collection is cycles / that time.
# Usage (GPU box): tools/pmc_issue.sh <out_dir> [first workload index] [how many]     ->  <out_dir>/<workload>/pass*/ + <out_dir>/summary.json (tools/issue_model.py)
set -u
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export VRT_BENCH_PREROLL=0
WORKLOADS=(
 "dragon primary 1920 1080"
 "dragon primary_shadow 1920 1080"
 "dragon full 1920 1080"
 "terrain primary 1920 1080"
 "monu9 primary 1280 720"
 "nature primary_shadow 3840 2160"
 "terrain_full primary 1920 1080"
 "terrain_full primary_shadow 1920 1080"
 "room full 1920 1080"
)
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU2 SQ_INST_LEVEL_VMEM SQ_IFETCH"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
FIRST=${1:-0}; COUNT=${2:-${#WORKLOADS[@]}}
for wl in "${WORKLOADS[@]:$FIRST:$COUNT}"; do
  set -- $wl
  d="$OUT/$1_$2_$3x$4"
  mkdir -p "$d"
  i=0
  for p in "${PASSES[@]}"; do
    rocprofv3 --pmc $p --kernel-trace --output-format csv -d "$d/pass$i" -- python3 "$REPO/bench.py" --no-cpu-baseline --no-configs --steps 10 --warmup 2 \
        --map "$1" --mode "$2" --width "$3" --height "$4" > "$d/pass$i.log" 2>&1 || echo "$wl pass $i ($p) failed" >> "$OUT/errors.log"
    i=$((i+1))
  done
  echo "done $wl" >> "$OUT/progress.log"
done
python3 "$REPO/tools/issue_model.py" --collect "$OUT" > "$OUT/summary.json" 2> "$OUT/summary.err"
tail -c 600 "$OUT/summary.json"
