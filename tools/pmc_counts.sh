#!/bin/bash
# One rocprofv3 --pmc pass (instruction counts and wave cycles) per kernel variant: tools/pmc_counts.sh <out_dir> <variant>...
set -u
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export VRT_BENCH_PREROLL=0   # counters are per launch: the untimed pre-roll frames would only add passes
for v in "$@"; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/v$v/pass0" -- python3 "$REPO/bench.py" --no-cpu-baseline --steps 10 --warmup 2 --variant "$v" ${PMC_BENCH_ARGS:-} > "$OUT/v$v.log" 2>&1 || echo "variant $v failed" >> "$OUT/errors.log"
  echo "== variant $v"; python3 "$REPO/tools/pmc_summary.py" "$OUT/v$v" | grep -E "trace_kernel|INSTS_VALU|INSTS_SALU|WAVE_CYCLES|ACTIVE_INST_VALU|insts / wave"
done
