#!/bin/bash
# rocprofv3 PMC passes for the full-shader kernel (mode 2) via tools/variant_sweep.py. Usage: tools/pmc_full.sh <out_dir> [sweep args]
set -u
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for p in "${PASSES[@]}"; do
  rocprofv3 --pmc $p --output-format csv -d "$OUT/pass$i" -- python3 "$REPO/tools/variant_sweep.py" --modes 2 --variants 0 --iters 10 "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i ($p) failed" >> "$OUT/errors.log"
  i=$((i+1))
done
python3 "$REPO/tools/pmc_summary.py" "$OUT" "trace_kernel<2" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
