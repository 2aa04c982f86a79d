#!/bin/bash
# Everything profiles/r02_* is made of, in one go on the GPU box: tools/r02_profile_all.sh <out_dir>
set -u
OUT=$(realpath -m "$1")
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd "$REPO"
python3 bench.py --steps 200 --warmup 20 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_n1_driver_shape.json" 2>> "$OUT/bench_n1.err"
VRT_BENCH_PREROLL=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_n1_driver_shape_no_preroll.json" 2>> "$OUT/bench_n1.err"
python3 bench.py --steps 200 --warmup 20 --extras --no-cpu-baseline > "$OUT/bench_n1_extras.json" 2>> "$OUT/bench_n1.err"
: > "$OUT/other_configs.jsonl"
for cfg in "--mode primary_shadow" "--mode full" "--map terrain" "--map terrain --mode primary_shadow" "--map monu9 --width 1280 --height 720" \
           "--map nature --width 3840 --height 2160 --mode primary_shadow"; do
  python3 bench.py --no-cpu-baseline $cfg >> "$OUT/other_configs.jsonl" 2>> "$OUT/bench_n1.err"
done
echo "bench lines done" 
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernel_trace" -- python3 "$REPO/bench.py" --no-cpu-baseline --steps 200 --warmup 20 > "$OUT/kernel_trace.log" 2>&1 )
find "$OUT/kernel_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
echo "kernel trace done"
bash tools/pmc.sh "$OUT/pmc_default" > "$OUT/pmc_default.log" 2>&1
echo "pmc default done"
PMC_BENCH_ARGS="--mode primary_shadow" bash tools/pmc_counts.sh "$OUT/counts_shadow" 0 > "$OUT/counts_shadow.txt" 2>&1
PMC_BENCH_ARGS="--mode full" bash tools/pmc_counts.sh "$OUT/counts_full" 0 > "$OUT/counts_full.txt" 2>&1
PMC_BENCH_ARGS="--map terrain" bash tools/pmc_counts.sh "$OUT/counts_terrain" 0 > "$OUT/counts_terrain.txt" 2>&1
echo "pmc counts done"
python3 bench.py --gpus 2 --backend gloo --steps 40 --warmup 8 > "$OUT/bench_gloo2_rehearsal.json" 2> "$OUT/bench_gloo2_rehearsal.err"
echo "all done"
