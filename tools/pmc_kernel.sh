#!/bin/bash
# rocprofv3 --pmc passes (memory-side and issue-side counters) for any command; prints per-kernel averages.
# usage: tools/pmc_kernel.sh <out_dir> <kernel-name-substring> -- <python script and args>
set -u
OUT=$(realpath -m "$1"); KERN="$2"; shift 3
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
)
i=0
for p in "${PASSES[@]}"; do
  rocprofv3 --pmc $p --output-format csv -d "$OUT/pass$i" -- python3 "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed" >> "$OUT/errors.log"
  i=$((i+1))
done
python3 "$REPO/tools/pmc_summary.py" "$OUT" "$KERN"
