#!/usr/bin/env python3
"""Builds the files bench.py quotes for its per-launch PMC figures, from what was collected on the GPU box:

    tools/issue_model.py --collect <dir written by tools/pmc_issue.sh>      -> JSON on stdout: per workload and kernel, the average of
                                                                             every counter and of the dispatch durations of each pass
    tools/issue_model.py --build profiles/r03_pmc_summary.json               -> profiles/r03_issue_model.json + profiles/pmc_traffic.json entries

The issue model (bench.py issue_roofline): per workload
    valu_insts_per_launch, salu_insts_per_launch   SQ_INSTS_VALU / SQ_INSTS_SALU of the ordered flavour of trace_kernel (the launches a
                                                   frame loop runs 15 times out of 16)
    busy_cycles                                    SQ_BUSY_CYCLES / 32 shader engines: the cycles the kernel's waves were on the chip
    clock_ghz                                      busy_cycles / the dispatch's own duration in the same pass (GRBM_GUI_ACTIVE also counts
                                                   the profiler's own packets around the dispatch and reads above the nominal clock)
    frac_under_pmc                                 valu_insts x issue_slot_cycles / (1024 SIMDs x busy_cycles): counters only, no clock
    half_rate_share                                static share of half- and quarter-rate kinds among the vector instructions of the
                                                   kernel's loops (tools/isa_cost.py on the build's assembly)
and globally, from tools/micro/valu_rate (profiles/r03_valu_rate.json, SIMDs that verifiably held w waves):
    issue_slot_cycles   what a saturated SIMD needs per wave64 vector instruction of a full-rate or mixed stream
    half_pipe_cycles    ... per half-rate instruction when only those are issued
    salu_cycles         ... per scalar instruction on the scalar port
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XCDS = 8
SHADER_ENGINES = 32   # 8 XCDs x 4: SQ_BUSY_CYCLES is summed over them


def collect(root):
    out = {}
    for wdir in sorted(glob.glob(os.path.join(root, "*_*x*"))):
        if not os.path.isdir(wdir):
            continue
        wl = os.path.basename(wdir)
        acc = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values
        dur = defaultdict(lambda: defaultdict(list))      # kernel -> pass -> ns
        for path in glob.glob(os.path.join(wdir, "pass*", "**", "*counter_collection.csv"), recursive=True):
            pas = re.search(r"(pass\d+)", path).group(1)
            seen = set()
            for row in csv.DictReader(open(path)):
                k = row.get("Kernel_Name", "?")
                if "trace_kernel" not in k and "denoise" not in k:
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                did = row.get("Dispatch_Id")
                if did not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    seen.add(did)
                    dur[k][pas].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
        # kernel-trace CSV of the same passes (durations when the counter CSV carries no timestamps)
        for path in glob.glob(os.path.join(wdir, "pass*", "**", "*kernel_trace.csv"), recursive=True):
            pas = re.search(r"(pass\d+)", path).group(1)
            for row in csv.DictReader(open(path)):
                k = row.get("Kernel_Name", "?")
                if "trace_kernel" not in k and "denoise" not in k:
                    continue
                if pas not in dur[k] or len(dur[k][pas]) == 0 or dur[k].get("_from_trace_" + pas):
                    dur[k]["_from_trace_" + pas] = True
                    dur[k].setdefault(pas + "_trace", []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
        kernels = {}
        for k, ctrs in acc.items():
            kernels[k] = {"counters": {n: sum(v) / len(v) for n, v in ctrs.items()}, "launches": max(len(v) for v in ctrs.values()),
                          "dispatch_ns_by_pass": {p: sum(v) / len(v) for p, v in dur[k].items() if isinstance(v, list) and v}}
        out[wl] = kernels
    return out


def probe_constants():
    t = json.load(open(os.path.join(ROOT, "profiles", "r03_valu_rate.json")))

    def best(name):
        v = t[name]
        ok = [(int(w[1:]), x["simd"]) for w, x in v.items() if x.get("simds_ok", 0) >= 256 and x["simd"] > 0]
        return min(s for _, s in ok), {f"w{w}": s for w, s in ok}
    full = ["v_add_f32", "v_mul_f32", "v_fma_f32", "v_sub_f32", "v_and_b32", "v_add_u32", "v_xor_b32", "v_mov_b32"]
    mixes = ["v_add_f32 + v_cndmask_b32 e64 (2 insts)", "3 x v_add_f32 + v_cndmask_b32 e64 (4 insts)", "v_and_b32 + v_add_f32 (2 insts)",
             "v_mul + v_floor + v_add + v_mul f32 (4 insts)", "v_mul_f32 + v_add_f32 (2 insts)"]
    half = ["v_cndmask_b32 e64 (fixed sgpr mask)", "v_cmp_lt_f32 -> sgpr pair", "v_floor_f32", "v_cvt_i32_f32", "v_lshl_add_u32", "v_bfe_u32",
            "v_min_f32", "v_max_f32", "v_cvt_f32_ubyte0", "v_lshlrev_b32"]
    salu = ["s_and_b64", "s_add_u32", "s_cselect_b32"]
    rows = {}
    for grp, names in (("full_rate", full), ("full_half_mixes_per_instruction", mixes), ("half_rate", half), ("scalar", salu)):
        rows[grp] = {n: best(n)[0] for n in names}
    slot = max(rows["full_rate"].values())   # the dearest full-rate kind at its best residency: what every vector instruction is charged
    return {"issue_slot_cycles": round(slot, 3), "half_pipe_cycles": round(sum(rows["half_rate"].values()) / len(rows["half_rate"]), 3),
            "salu_cycles": round(sum(rows["scalar"].values()) / len(rows["scalar"]), 3), "probe_rows": rows}


def half_share(mode):
    """static share of half/quarter-rate kinds among the vector instructions inside the loops of the kernel a workload of `mode` ran
    (mode "full_opaque": the stack-free full path tracer, trace_kernel<6, ...>)"""
    src = {"primary": "vrt_launch_primary.hip", "primary_shadow": "vrt_launch_shadow.hip", "full": "vrt_launch_full.hip", "full_opaque": "vrt_launch_full.hip"}[mode]
    sym = {"primary": "trace_kernelILi0ENS_2v45TravTILb1EEELi8ELi64ELi7ELb0ELi1EEE", "primary_shadow": "trace_kernelILi1ENS_2v45TravTILb1EEELi8ELi64ELi7ELb0ELi1EEE",
           "full": "trace_kernelILi2ENS_2v45TravTILb0EEELi8ELi64ELi5ELb0ELi1EEE", "full_opaque": "trace_kernelILi6ENS_2v45TravTILb1EEELi8ELi64ELi6ELb0ELi1EEE"}[mode]
    csrc = os.path.join(ROOT, "voxel-raytracer_amd", "csrc")
    asm = f"/tmp/issue_model_{mode}.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                           "--cuda-device-only", "-S", "-o", asm, os.path.join(csrc, src)], stderr=subprocess.DEVNULL)
    txt = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "isa_cost.py"), asm, sym, "--blocks"], text=True)
    cnt = defaultdict(int)
    for line in txt.splitlines():
        if "in Loop" not in line and "Loop Header" not in line and "Parent Loop" not in line:
            continue
        for k, v in re.findall(r"(valu-full|valu-half|valu-quarter|v_cmp):(\d+)", line):
            cnt[k] += int(v)
    tot = sum(cnt.values())
    return round((cnt["valu-half"] + cnt["valu-quarter"] + cnt["v_cmp"]) / max(tot, 1), 3), dict(cnt)


def build(summary_path):
    sys.path.insert(0, ROOT)
    import bench
    import vrt_import
    S = json.load(open(summary_path))
    consts = probe_constants()
    model = {"_note": "see tools/issue_model.py; inputs: " + os.path.relpath(summary_path, ROOT) + ", profiles/r03_valu_rate.json",
             "simds": 1024, "lib_stamp": bench.lib_stamp(vrt_import.vrt()), **{k: consts[k] for k in ("issue_slot_cycles", "half_pipe_cycles", "salu_cycles")},
             "probe_rows": consts["probe_rows"], "workloads": {}}
    traffic_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic = json.load(open(traffic_path))
    shares = {}
    for wl, kernels in S.items():
        m = re.match(r"(.+)_(primary_shadow|primary|full)_(\d+)x(\d+)$", wl)
        mp, mode, W, H = m.group(1), m.group(2), int(m.group(3)), int(m.group(4))
        # the ordered flavour (SCHED = 1: "Lb0ELi1EEE") is what a frame loop runs; fall back to the plain one
        pick = None
        for k in kernels:
            if "trace_kernel" in k and (", 1>" in k or "Lb0ELi1E" in k):
                pick = k
        if pick is None:
            pick = next((k for k in kernels if "trace_kernel" in k and (", 0>" in k or "Lb0ELi0E" in k)), None)
        if pick is None:
            continue
        c = kernels[pick]["counters"]
        d = kernels[pick]["dispatch_ns_by_pass"]
        kmode = mode
        if mode == "full" and "trace_kernel<6" in pick:
            kmode = "full_opaque"
        if kmode not in shares:
            shares[kmode] = half_share(kmode)
        key = f"{mp}/{W}x{H}/{mode}/variant0"
        e = {"kernel": pick, "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "salu_insts_per_launch": c.get("SQ_INSTS_SALU"),
             "waves": c.get("SQ_WAVES"), "trans_insts": c.get("SQ_INSTS_VALU_TRANS_F32"), "cvt_insts": c.get("SQ_INSTS_VALU_CVT"),
             "half_rate_share": shares[kmode][0], "half_rate_share_from": {"static loop blocks": shares[kmode][1]}}
        # the kernel's own cycles: SQ_BUSY_CYCLES (cycles with waves present, summed over the shader engines); its duration in the
        # same pass gives the clock it ran at under the collection
        ns0 = d.get("pass0") or d.get("pass0_trace")
        if c.get("SQ_BUSY_CYCLES") and e["valu_insts_per_launch"]:
            busy = c["SQ_BUSY_CYCLES"] / SHADER_ENGINES
            e["busy_cycles"] = round(busy, 1)
            e["frac_under_pmc"] = round(e["valu_insts_per_launch"] * consts["issue_slot_cycles"] / 1024 / busy, 4)
            e["frac_at_2_cycles_under_pmc"] = round(e["valu_insts_per_launch"] * 2.0 / 1024 / busy, 4)
            e["scalar_port_frac_under_pmc"] = round((e["salu_insts_per_launch"] or 0) * consts["salu_cycles"] / 1024 / busy, 4)
            e["half_pipe_frac_under_pmc"] = round(e["valu_insts_per_launch"] * shares[kmode][0] * consts["half_pipe_cycles"] / 1024 / busy, 4)
            if ns0:
                e["kernel_ms_under_pmc"] = round(ns0 * 1e-6, 5)
                e["clock_ghz"] = round(busy / ns0, 4)
        if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAVES"):
            e["wave_cycles_per_wave"] = round(c["SQ_WAVE_CYCLES"] * 4 / c["SQ_WAVES"], 1)
        model["workloads"][key] = e
        if c.get("FETCH_SIZE") is not None and c.get("WRITE_SIZE") is not None:
            traffic[key + "/gpus1"] = {"bytes": int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), "fetch_kb": round(c["FETCH_SIZE"], 1),
                                       "write_kb": round(c["WRITE_SIZE"], 1), "round": 3,
                                       "source": f"{os.path.relpath(summary_path, ROOT)} [{wl}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                                                 "tools/pmc_issue.sh; FETCH_SIZE not doubled: 8-byte gathers, not wide streams)"}
    json.dump(model, open(os.path.join(ROOT, "profiles", "r03_issue_model.json"), "w"), indent=1)
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    for k, e in model["workloads"].items():
        print(k, {x: e.get(x) for x in ("valu_insts_per_launch", "busy_cycles", "clock_ghz", "half_rate_share", "kernel_ms_under_pmc", "frac_under_pmc",
                                        "frac_at_2_cycles_under_pmc", "scalar_port_frac_under_pmc", "half_pipe_frac_under_pmc")})


if __name__ == "__main__":
    if sys.argv[1] == "--collect":
        json.dump(collect(sys.argv[2]), sys.stdout, indent=1)
    elif sys.argv[1] == "--build":
        build(sys.argv[2])
