#!/bin/bash
# rocprofv3 PMC passes for the display-pass kernels (tools/denoise_time.py). Usage: tools/pmc_denoise.sh <out_dir> [map W H]
set -u
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
i=0
for p in "${PASSES[@]}"; do
  DENOISE_ITERS=5 rocprofv3 --pmc $p --output-format csv -d "$OUT/pass$i" -- python3 "$REPO/tools/denoise_time.py" "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i ($p) failed" >> "$OUT/errors.log"
  i=$((i+1))
done
python3 "$REPO/tools/pmc_summary.py" "$OUT" denoise > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
