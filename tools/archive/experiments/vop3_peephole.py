#!/usr/bin/env python3
"""ISA peephole for the traversal kernels: re-encode the full-rate 32-bit VALU opcodes from VOP2 (_e32) to VOP3 (_e64).

Measured on gfx950 at the kernels' occupancy (tools/gen_valu_probe.py -> tools/valu_probe3.hip, profiles/r03_valu_probe3.txt): on
lane-varying operands v_mul_f32 / v_add_f32 / v_sub_f32 / v_add_u32 / v_sub_u32 / v_and_b32 / v_lshrrev_b32 cost 3.8-3.9 cycles per
wave64 instruction in their VOP2 encoding and 2.8-3.0 in VOP3; v_fmac_f32_e32 costs 3.8 where v_fma_f32 with the accumulator as third
source costs 2.9.  The compiler always picks the shorter VOP2 encoding.  Same opcode, same operands, same result bits -- only the
encoding (8 bytes instead of 4) changes, so this is not a numerical change of any kind.

Only instructions whose operands VOP3 can encode on gfx9 are touched: registers and inline constants, no 32-bit literals (VOP3 has no
literal slot before gfx10), no SDWA / DPP forms.  Reads assembly on argv[1], writes argv[2], prints the conversion counts.
"""
import re
import sys

PLAIN = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
         "v_lshrrev_b32")
INLINE_FLOATS = {"0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0"}
REG = re.compile(r"^(v\d+|s\d+|vcc_lo|vcc_hi|m0|exec_lo|exec_hi|ttmp\d+)$")


def encodable(tok):
    tok = tok.strip()
    if REG.match(tok):
        return True
    if tok in INLINE_FLOATS or tok == "0":
        return True
    if re.match(r"^-?\d+$", tok):
        return -16 <= int(tok) <= 64
    return False  # hex literals, symbols, anything else: leave the instruction alone


ONLY_CNDMASK = len(sys.argv) > 3 and sys.argv[3] == "cndmask"   # second experiment: only v_cndmask_b32_e32 ..., vcc -> _e64 (22 cycles in the probe)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    counts = {}
    out = []
    pat = re.compile(r"^(\s+)(v_[a-z0-9_]+?)_e32\s+(.*?)(\s*;.*)?$")
    for line in open(src):
        m = pat.match(line.rstrip("\n"))
        if m:
            indent, op, operands, comment = m.group(1), m.group(2), m.group(3), m.group(4) or ""
            toks = [t.strip() for t in operands.split(",")]
            if op in PLAIN and not ONLY_CNDMASK and len(toks) == 3 and all(encodable(t) for t in toks):
                out.append(f"{indent}{op}_e64 {', '.join(toks)}{comment}\n")
                counts[op] = counts.get(op, 0) + 1
                continue
            if op == "v_cndmask_b32" and ONLY_CNDMASK and len(toks) == 4 and toks[3] == "vcc" and all(encodable(t) for t in toks[:3]):
                out.append(f"{indent}v_cndmask_b32_e64 {', '.join(toks)}{comment}\n")
                counts[op] = counts.get(op, 0) + 1
                continue
            if ONLY_CNDMASK:
                out.append(line)
                continue
            if op == "v_fmac_f32" and len(toks) == 3 and all(encodable(t) for t in toks):
                out.append(f"{indent}v_fma_f32 {toks[0]}, {toks[1]}, {toks[2]}, {toks[0]}{comment}\n")
                counts[op] = counts.get(op, 0) + 1
                continue
        out.append(line)
    open(dst, "w").writelines(out)
    print("vop3_peephole:", ", ".join(f"{k} {v}" for k, v in sorted(counts.items())) or "nothing to convert")


if __name__ == "__main__":
    main()
