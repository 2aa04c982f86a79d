#!/bin/bash
# Collects rocprofv3 PMC counters for bench.py in separate passes (kernel-trace/--stats are never combined
# with --pmc here). Usage (on the GPU box): tools/pmc.sh <out_dir> [bench args...]
set -u
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export VRT_BENCH_PREROLL=0   # counters are per launch: the untimed pre-roll frames would only add passes
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SENDMSG SQ_ACTIVE_INST_FLAT"
 "SQ_INSTS_WAVE32 SQ_WAVES_RESTORED SQ_WAVE_READY SQ_WAVE_DEP_WAIT SQ_WAVE_ISSUE_WAIT SQ_WAVE_SCHED_WAIT SQ_IFETCH_LEVEL SQ_ACCUM_PREV"
)
i=0
for p in "${PASSES[@]}"; do
  rocprofv3 --pmc $p --output-format csv -d "$OUT/pass$i" -- python3 "$REPO/bench.py" --no-cpu-baseline --steps 10 --warmup 2 "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i ($p) failed" >> "$OUT/errors.log"
  i=$((i+1))
done
python3 "$REPO/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
