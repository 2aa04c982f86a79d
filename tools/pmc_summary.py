#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter CSVs per kernel: tools/pmc_summary.py <dir with pass*/ sub-dirs> [kernel name substring]."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "trace_kernel"
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row.get("Kernel_Name", "?")
            if want not in k:
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, ctrs in acc.items():
        print(k)
        avg = {n: sum(v) / len(v) for n, v in ctrs.items()}
        for n in sorted(avg):
            print(f"  {n:34s} {avg[n]:18.1f}   (n={len(ctrs[n])})")
        g = avg.get
        if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
            print(f"  -> VALU insts / wave              {g('SQ_INSTS_VALU') / g('SQ_WAVES'):12.1f}")
        if g("SQ_INSTS_VMEM_RD") and g("SQ_WAVES"):
            print(f"  -> VMEM reads / wave              {g('SQ_INSTS_VMEM_RD') / g('SQ_WAVES'):12.1f}")
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            print(f"  -> avg active lanes per VALU inst {g('SQ_THREAD_CYCLES_VALU') / g('SQ_ACTIVE_INST_VALU') / 4 * 64 / 16:12.2f} (uncalibrated)")
        if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
            print(f"  -> wait_any / wave_cycles         {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):12.3f}")
        if g("SQ_WAIT_INST_ANY") and g("SQ_WAVE_CYCLES"):
            print(f"  -> wait_inst_any / wave_cycles    {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):12.3f}")
        if g("SQ_ACTIVE_INST_ANY") and g("SQ_WAVE_CYCLES"):
            print(f"  -> active_inst_any / wave_cycles  {g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES'):12.3f}")
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
            print(f"  -> L2 hit rate                    {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):12.4f}")
        if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
            print(f"  -> L1 miss ratio (TCC reads/TCP acc) {g('TCP_TCC_READ_REQ_sum') / g('TCP_TOTAL_CACHE_ACCESSES_sum'):9.4f}")
        if g("FETCH_SIZE") is not None:
            print(f"  -> FETCH_SIZE KB (x2 correction for wide streams NOT applied) {g('FETCH_SIZE'):12.1f}")
        if g("WRITE_SIZE") is not None:
            print(f"  -> WRITE_SIZE KB {g('WRITE_SIZE'):12.1f}")


if __name__ == "__main__":
    main()
