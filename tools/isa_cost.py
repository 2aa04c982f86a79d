#!/usr/bin/env python3
"""Static issue-cost estimate of a gfx950 kernel from its assembly and the measured per-instruction SIMD costs.

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -o k.s one.hip
    python3 tools/isa_cost.py k.s <kernel-name-substring> [--table profiles/r02_valu_rate.json] [--w w6] [--blocks]

For every basic block of the kernel: instructions by class and the SIMD time they cost when the SIMD is saturated
(tools/micro/valu_rate: ticks per instruction per SIMD at w resident waves). The traversal kernels are issue bound, so the
sum over a loop body, weighted by trip counts, is the time a DDA step costs; use it to compare two builds of a loop
without a GPU. Unknown mnemonics are priced by encoding class (VOP3 integer 4.4, other VALU 2.6).
Vector and scalar instructions issue from different ports (a v_add_f32 + s_add_u32 pair costs what the slower of the two
does), so the totals are printed per port; the vector port is the one that binds these kernels.
"""
import argparse
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# mnemonic (without _e32/_e64/_sdwa/_dpp) -> row of the valu_rate table
MAP = {
    "v_add_f32": "v_add_f32", "v_sub_f32": "v_sub_f32", "v_subrev_f32": "v_sub_f32", "v_mul_f32": "v_mul_f32",
    "v_fma_f32": "v_fma_f32", "v_fmac_f32": "v_fma_f32", "v_mac_f32": "v_fma_f32", "v_mad_f32": "v_fma_f32",
    "v_pk_mul_f32": "v_pk_mul_f32", "v_pk_add_f32": "v_pk_add_f32", "v_pk_fma_f32": "v_pk_fma_f32",
    "v_cndmask_b32": "v_cndmask_b32 e64 (fixed sgpr mask)",
    "v_and_b32": "v_and_b32", "v_or_b32": "v_and_b32", "v_xor_b32": "v_xor_b32", "v_not_b32": "v_and_b32",
    "v_lshlrev_b32": "v_lshlrev_b32", "v_lshrrev_b32": "v_lshrrev_b32 by vgpr", "v_ashrrev_i32": "v_lshrrev_b32 by vgpr",
    "v_bfe_u32": "v_bfe_u32", "v_bfe_i32": "v_bfe_u32", "v_bfi_b32": "v_bfi_b32", "v_perm_b32": "v_perm_b32",
    "v_add_u32": "v_add_u32", "v_sub_u32": "v_sub_u32", "v_subrev_u32": "v_sub_u32", "v_add_co_u32": "v_add_u32",
    "v_addc_co_u32": "v_add_u32", "v_sub_co_u32": "v_sub_u32", "v_subb_co_u32": "v_sub_u32",
    "v_lshl_add_u32": "v_lshl_add_u32", "v_add_lshl_u32": "v_add_lshl_u32", "v_and_or_b32": "v_and_or_b32",
    "v_or3_b32": "v_or3_b32", "v_lshl_or_b32": "v_lshl_or_b32", "v_add3_u32": "v_add3_u32", "v_xad_u32": "v_add3_u32",
    "v_cvt_flr_i32_f32": "v_cvt_flr_i32_f32", "v_cvt_f32_i32": "v_cvt_f32_i32", "v_cvt_f32_u32": "v_cvt_f32_i32",
    "v_cvt_f32_ubyte0": "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte1": "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte2": "v_cvt_f32_ubyte0",
    "v_cvt_f32_ubyte3": "v_cvt_f32_ubyte0", "v_cvt_u32_f32": "v_cvt_u32_f32", "v_cvt_i32_f32": "v_cvt_u32_f32",
    "v_rndne_f32": "v_floor_f32", "v_trunc_f32": "v_floor_f32", "v_ceil_f32": "v_floor_f32",
    "v_rcp_f32": "v_rcp_f32", "v_sqrt_f32": "v_sqrt_f32", "v_rsq_f32": "v_rsq_f32", "v_rcp_iflag_f32": "v_rcp_f32",
    "v_div_scale_f32": "v_div_scale_f32", "v_div_fmas_f32": "v_div_fmas_f32", "v_div_fixup_f32": "v_div_fixup_f32",
    "v_mad_u64_u32": "v_mad_u64_u32", "v_lshl_add_u64": "v_mad_u64_u32", "v_lshlrev_b64": "v_lshlrev_b64",
    "v_mov_b32": "v_mov_b32", "v_mov_b64": "v_mov_b32", "v_readfirstlane_b32": "v_readfirstlane_b32",
    "v_readlane_b32": "v_readfirstlane_b32", "v_writelane_b32": "v_readfirstlane_b32",
    "v_min_f32": "v_min_f32", "v_max_f32": "v_max_f32", "v_med3_f32": "v_med3_f32", "v_min3_f32": "v_min3_f32",
    "v_max3_f32": "v_min3_f32", "v_floor_f32": "v_floor_f32", "v_fract_f32": "v_fract_f32", "v_ldexp_f32": "v_ldexp_f32",
    "v_mad_u32_u24": "v_mad_u32_u24", "v_mul_u32_u24": "v_mul_u32_u24", "v_mul_lo_u32": "v_mul_lo_u32",
    "v_mul_hi_u32": "v_mul_lo_u32", "v_max_u32": "v_max_u32", "v_min_u32": "v_min_u32", "v_max_i32": "v_max_u32",
    "v_min_i32": "v_min_u32",
}
SUFFIX = re.compile(r"_(e32|e64|sdwa|dpp)$")


def load_table(path, w):
    """SIMD ticks per instruction at w resident waves (tools/micro/valu_rate: measured on SIMDs that verifiably held w waves);
    where no SIMD qualified at that w, the largest w that has a figure."""
    t = json.load(open(path))
    out = {}
    for k, v in t.items():
        for ww in [w] + sorted(v, key=lambda x: -int(x[1:])):
            if v.get(ww, {}).get("simd", 0) > 0:
                out[k] = v[ww]["simd"]
                break
    return out


def price(mn, table):
    base = SUFFIX.sub("", mn)
    if base.startswith("v_cmp"):
        return table["v_cmp_lt_f32 -> sgpr pair"], "v_cmp"
    if base.startswith("s_cbranch") or base == "s_branch":
        return table["s_cbranch_execz not taken"], "branch"
    if base.startswith("s_waitcnt") or base == "s_nop":
        return table["s_and_b64"], "salu"
    if base.startswith("s_load") or base.startswith("s_buffer_load"):
        return table["s_and_b64"], "smem"
    if base.startswith("s_"):
        return table["s_and_b64"], "salu"
    if base.startswith(("global_", "buffer_", "flat_", "scratch_", "ds_")):
        return 4.0, "mem"   # issue slot only; the TA / LDS time is not an issue cost
    if base in MAP and MAP[base] in table:
        c = table[MAP[base]]
        cls = "valu-full" if c < 3.2 else ("valu-half" if c < 6.0 else "valu-quarter")
        return c, cls
    if base.startswith("v_"):
        return (4.2, "valu-half?") if mn.endswith("_e64") or re.search(r"3_|_or_|_add_|lshl", base) else (2.3, "valu-full?")
    return 0.0, "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--table", default=os.path.join(ROOT, "profiles", "r03_valu_rate.json"))
    ap.add_argument("--w", default="w8")
    ap.add_argument("--blocks", action="store_true", help="per basic block lines")
    ap.add_argument("--top", type=int, default=0, help="the N most expensive mnemonics of the whole kernel")
    a = ap.parse_args()
    table = load_table(a.table, a.w)
    lines = open(a.asm).read().split("\n")
    start = next((i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w.$]*:", l) and a.kernel in l and not l.startswith(".L")), None)
    if start is None:
        sys.exit("kernel not found")
    blocks = [("entry", start + 1, [])]
    unknown = collections.Counter()
    total = collections.Counter()
    cost_by_mn = collections.Counter()
    for i in range(start + 1, len(lines)):
        l = lines[i]
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            blocks.append((m.group(1) + ("  " + m.group(2).strip() if "Loop" in m.group(2) else ""), i + 1, []))
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\b", l)
        if not m or l.strip().startswith((";", ".")):
            continue
        mn = m.group(1)
        c, cls = price(mn, table)
        if cls.endswith("?"):
            unknown[mn] += 1
        blocks[-1][2].append((mn, c, cls))
        total[cls.rstrip("?")] += 1
        cost_by_mn[SUFFIX.sub("", mn)] += c
    grand = 0.0
    for name, line_no, ins in blocks:
        c = sum(x[1] for x in ins)
        cv = sum(x[1] for x in ins if x[2].startswith(("valu", "v_cmp")))
        grand += c
        if a.blocks and ins:
            by = collections.Counter(x[2].rstrip("?") for x in ins)
            print(f"{name:<60s} L{line_no - start:<5d} n={len(ins):3d} vector={cv:6.1f} other={c - cv:6.1f}  " + " ".join(f"{k}:{v}" for k, v in sorted(by.items())))
    n = sum(total.values())
    print(f"kernel total: {n} instructions, static SIMD cost {grand:.0f} ticks ({grand / max(n, 1):.2f} per instruction at {a.w}); by class: "
          + ", ".join(f"{k} {v}" for k, v in sorted(total.items())))
    if unknown:
        print("priced by encoding class (not in the table):", dict(unknown))
    if a.top:
        for mn, c in cost_by_mn.most_common(a.top):
            print(f"  {mn:<24s} {c:8.1f}")


if __name__ == "__main__":
    main()
