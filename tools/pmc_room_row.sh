#!/bin/bash
# counters of the room's heaviest tile row alone. Usage: tools/pmc_room_row.sh <out_dir>
set -u
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM"
 "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM SQ_INSTS_BRANCH"
)
i=0
for p in "${PASSES[@]}"; do
  rocprofv3 --pmc $p --output-format csv -d "$OUT/pass$i" -- python3 "$REPO/tools/room_row.py" > "$OUT/pass$i.log" 2>&1 || echo "pass $i ($p) failed" >> "$OUT/errors.log"
  i=$((i+1))
done
python3 "$REPO/tools/pmc_summary.py" "$OUT" "trace_kernel<2" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
