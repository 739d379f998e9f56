"""Prices the vector instructions of a gfx950 assembly listing (hipcc -S) with the issue rates measured by
scripts/valu_rate_probe.hip at eight waves per SIMD: plain f32 add / sub / mul / fma and the simple bitwise, add and
right-shift integer operations on registers or constants retire at about one per 2.3 cycles per SIMD, every other
vector instruction (anything reading an SGPR, conversions, min / max, compares, selects, three-operand integer forms,
packed f32) at one per 4, the transcendental unit's at one per 8.

    python scripts/isa_issue_cost.py listing.s [kernel-name-substring] [first-label last-label]
"""
import collections
import re
import sys

FAST = {"v_fma_f32", "v_fmaak_f32", "v_fmamk_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
CYCLES = {"fast": 2.3, "normal": 4.0, "trans": 8.0}


def classify(line):
    parts = line.split(None, 1)
    op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", parts[0])
    if op in TRANS:
        return "trans"
    operands = parts[1] if len(parts) > 1 else ""
    if parts[0].endswith("_dpp") or parts[0].endswith("_sdwa") or "quad_perm" in operands or "row_" in operands:
        return "normal"
    sources = operands.split(",")[1:]
    reads_scalar = any(re.match(r"\s*-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec|m0)", s) for s in sources)
    if op in FAST and not reads_scalar:
        return "fast"
    return "normal"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    first, last = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else (None, None)
    kernel, active = None, first is None
    totals = collections.defaultdict(collections.Counter)
    for raw in open(path):
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            kernel, active = m.group(1), first is None
            continue
        if kernel is None or want not in kernel:
            continue
        label = re.match(r"^(\.LBB\w+):", raw)
        if label and first is not None:
            if label.group(1) == first:
                active = True
            elif label.group(1) == last:
                active = False
        text = raw.strip()
        if active and text.startswith("v_"):
            totals[kernel][classify(text)] += 1
    for kernel, c in totals.items():
        n = sum(c.values())
        cycles = sum(CYCLES[k] * v for k, v in c.items())
        print(f"{kernel[:90]}\n   {n} vector instructions: {dict(c)}; {cycles:.0f} issue cycles = {cycles / n:.2f} per instruction")


if __name__ == "__main__":
    main()
