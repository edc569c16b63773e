#!/usr/bin/env python3
"""The opcode mix of the dominant trace kernel and the VALU-issue ceiling that mix can reach.

bench.py prices the kernel against the guide's VALU issue peak (one wave64 VALU instruction per SIMD per 2 cycles:
256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1228.8 G wave-instructions / s).  No instruction this kernel is made of issues that fast
(profiles/r02_valu_probe.txt, measured per opcode at the kernel's occupancy), so next to `frac` the line carries `mix_ceiling`: the
rate a SIMD would reach if it issued THIS kernel's dynamic opcode mix back to back,

    mix_ceiling = 1024 SIMDs x 2.4 GHz / (sum over opcodes of count x cycles) x (sum of counts)

with the dynamic histogram built here from three committed inputs:
  * the kernel's ISA: rc_traverse.hip compiled to assembly with -DRC_PHASE_MARKERS (comment lines delimit the interior / leaf /
    switch / finish phases; the marker build must have the same VALU instruction count and register use as the product build -- checked),
    VALU opcodes counted per phase;
  * how often a wave runs each phase: the pass counters of the kernel's STATS instantiation (tools/phase_passes.py on the GPU box,
    profiles/r06_phase_passes_kernel5.json);
  * cycles per opcode: profiles/r02_valu_probe.txt (tools/archive/valu_probe.hip), ns per wave-instruction per SIMD x 2.4 GHz.
The prediction sum(passes x static count) is compared with the measured SQ_INSTS_VALU of the counter file: that is the check that the
histogram describes what ran.

    python3 tools/isa_mix.py [--workload c3] > profiles/r06_isa_mix_kernel5.json
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raycore.jl_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize"] + os.environ.get("RC_EXTRA_FLAGS", "").split() + [ "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "--cuda-device-only", "-S"]
KERNEL = "k_trace_phased_ldsILb0ELi768ELi16ELi6ELb0ELb0ELb1EE"  # <closest, 768 threads, 16-entry LDS stacks, 6 waves / SIMD, no timeline, no stats, 16-bit stack entries (C3's trees are small)>
CLOCK_GHZ = 2.4


def assembly(markers):
    out = f"/tmp/rc_traverse_{'marked' if markers else 'plain'}.s"
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + (["-DRC_PHASE_MARKERS"] if markers else []) + [os.path.join(CSRC, "rc_traverse.hip"), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if KERNEL in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\S+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    meta = {}
    for l in lines[end:end + 40]:
        m = re.match(r"\s*\.set \S+\.(num_vgpr|numbered_sgpr|private_seg_size), (\d+)", l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    return lines[start + 1:end], meta


def instructions(body):
    """(region, mnemonic) for every instruction line; region follows the RC_MARK comment lines (textual order).  Basic blocks that a wave
    only enters on a rare path are reported as region + "_cold" and never weighted: the lane stack's global spill path (its address
    arithmetic is the only user of v_mad_u64_u32 / v_mul_hi_u32 in the phases) and the generic IEEE division behind safe_inv3's
    fast path (v_div_scale in the switch / finish phases; the leaf phase's 1 / det division is hot)."""
    region, block, blocks = "other", [], []
    for l in body:
        t = l.strip()
        m = re.match(r";\s*RC_MARK (\w+)_(begin|end)", t)
        if m:
            blocks.append((region, block)); block = []
            region = m.group(1) if m.group(2) == "begin" else "other"
            continue
        if t.endswith(":") and not t.startswith(";"):
            blocks.append((region, block)); block = []
            continue
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        block.append(op)
        if op.startswith("s_cbranch") or op == "s_branch":
            blocks.append((region, block)); block = []
    blocks.append((region, block))
    for region, ops in blocks:
        cold = region != "other" and (any(o in ("v_mad_u64_u32", "v_mul_hi_u32") for o in ops) or
                                      (region in ("entry", "refill") and any(o == "v_div_scale_f32" for o in ops)))
        for o in ops:
            yield (region + "_cold" if cold else region), o


def probe_cycles():
    """cycles per wave-instruction per SIMD from profiles/r02_valu_probe.txt"""
    table = {}
    for l in open(os.path.join(ROOT, "profiles", "r02_valu_probe.txt")):
        m = re.match(r"(.+?)\s+[\d.]+ ms\s+([\d.]+) ns per wave-instruction per SIMD", l)
        if m:
            table[m.group(1).strip()] = float(m.group(2)) * CLOCK_GHZ
    return table


def cycles_of(op, probe):
    """Measured cycles of the probe entry closest to `op` (destination != source where the probe has that form: compiled code rarely
    overwrites an operand); None when nothing comparable was probed."""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    e64 = op.endswith("_e64")
    if base in ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"):
        return probe.get(base + " d!=s", probe[base]) if base == "v_pk_mul_f32" else probe[base]
    if base in ("v_minimum3_f32", "v_maximum3_f32"):
        return probe["v_maximum3 d!=s"]
    if base in ("v_fma_f32", "v_fmac_f32"):
        return probe["v_fma_f32"]
    if base == "v_rcp_f32":
        return probe["v_rcp_f32"]
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        return probe["v_cmp_lt_f32_e64 sgpr"] if e64 else probe["v_cmp_lt_f32"]
    if base == "v_cndmask_b32":
        return probe["v_cndmask_b32_e64 sgpr"]
    if base == "v_mov_b32":
        return probe["v_mov_b32_e64"] if e64 else probe["v_mov_b32"]
    if base in ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32"):
        return probe["v_mul_f32_e64 d!=s"] if e64 else probe["v_mul_f32 d!=s"]
    if base in ("v_max_f32", "v_min_f32", "v_med3_f32", "v_max3_f32", "v_min3_f32"):
        return probe["v_max_f32_e64"]
    if base in ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32"):
        return probe["v_and_b32"] if not e64 else probe["v_mul_f32_e64 d!=s"]
    if base in ("v_lshl_add_u32", "v_add_lshl_u32", "v_add3_u32", "v_lshl_or_b32", "v_and_or_b32", "v_or3_b32", "v_bfe_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_lshlrev_b64",
                "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24"):
        return probe["v_lshl_add_u32"]
    if base == "v_mad_u64_u32":
        return probe["v_mad_u64_u32"]
    if base == "v_swap_b32":
        return probe["v_swap_b32"]
    return None


def probe3_tables():
    """{opcode form: (cycles on static operands, cycles on lane-varying operands)} from profiles/r03_valu_probe3.txt (tools/archive/gen_valu_probe.py:
    physical registers pinned, destination separate from the sources, 6 waves / SIMD)."""
    table = {}
    path = os.path.join(ROOT, "profiles", "r03_valu_probe3.txt")
    if not os.path.exists(path):
        return table
    for l in open(path):
        m = re.match(r"(\S+(?: \S+)?)\s+static ([\d.]+) ns = [\d.]+ cycles\s+varied ([\d.]+) ns", l)
        if m:
            table[m.group(1)] = (float(m.group(2)) * CLOCK_GHZ, float(m.group(3)) * CLOCK_GHZ)
    return table


def cycles3(op, table, which):
    """Price `op` with the pinned-register probe; `which` 0 = static operands, 1 = varied.  Opcodes the probe does not list are priced as
    the quarter-rate class they almost all belong to (every 3-source, 64-bit, shift-left, compare, select and min / max form measured
    4.5-4.7 cycles)."""
    if op in table:
        return table[op][which]
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    e64 = op.endswith("_e64")
    for cand in (base, base + "_e32", base + "_e64"):
        if cand in table and not (e64 and cand.endswith("_e32") and base + "_e64" in table):
            return table[cand][which]
    if base.startswith("v_cmp"):
        key = ("v_cmp_f32" if "_f32" in base else "v_cmp_u32") + ("_e64" if e64 else "_e32")
        return table[key][which]
    if base in ("v_subrev_f32",):
        return table["v_sub_f32" + ("_e64" if e64 else "_e32")][which]
    if base in ("v_subrev_u32", "v_or_b32", "v_xor_b32"):
        return table["v_and_b32" + ("_e64" if e64 else "_e32")][which]
    return table["v_lshl_add_u32"][which]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--passes", default=os.path.join(ROOT, "profiles", "r06_phase_passes_kernel5.json"))
    ap.add_argument("--counters", default=os.path.join(ROOT, "profiles", "r06_pmc_c3.json"))
    args = ap.parse_args()
    marked, meta_m = assembly(True)
    plain, meta_p = assembly(False)
    probe = probe_cycles()
    static = collections.defaultdict(collections.Counter)
    for region, op in instructions(marked):
        if op.startswith("v_"):
            static[region][op] += 1
    n_valu_marked = sum(sum(c.values()) for c in static.values())
    n_valu_plain = sum(1 for _, op in instructions(plain) if op.startswith("v_"))
    out = {"kernel": "k_trace_phased_lds<false, 768, 16, 6, false, false, true>", "source": "hipcc -S --cuda-device-only of raycore.jl_amd/csrc/rc_traverse.hip with the Makefile's flags",
           "marker_build_matches_product": {"valu_instructions": [n_valu_marked, n_valu_plain], "registers": [meta_m, meta_p],
                                            "same": n_valu_marked == n_valu_plain and meta_m == meta_p},
           "static_valu_instructions_per_phase": {r: sum(c.values()) for r, c in static.items()},
           "static_histogram_per_phase": {r: dict(c.most_common()) for r, c in static.items()}}
    passes = None
    if os.path.exists(args.passes):
        passes = json.load(open(args.passes))["workloads"][args.workload]
    if passes:
        # a wave executes a phase's instructions once per pass in which at least one of its lanes needs the phase (the compiler skips the
        # block on an empty EXEC mask); "other" = loop control and votes around the phases, once per outer iteration, and the interior
        # loop's own control, once per interior iteration
        # switch = the return to the top level (passes with a lane on the sentinel), entry = the instance entry (passes with a lane on a
        # TLAS leaf); finish = the refill pass's control, writeout / refill = its two bodies
        # the control around writeout / refill ("finish": votes, mbcnt ranks, the claim) and "other" are not weighted
        weight = {"interior": passes["interior_passes"], "leaf": passes["leaf_passes"], "switch": passes["exit_passes"], "entry": passes["entry_passes"],
                  "writeout": passes["writeout_passes"], "refill": passes["refill_rounds"]}
        dyn = collections.Counter()
        per_phase = {}
        for r, w in weight.items():
            per_phase[r] = w * sum(static[r].values())
            for op, c in static[r].items():
                dyn[op] += w * c
        # "other" (prologue, LDS staging, loop control and votes between the phases) is not weighted: what the phases do not explain
        # shows up as the residual against the measured SQ_INSTS_VALU below and is priced at the phases' average cycles
        total = sum(dyn.values())
        cyc, unprobed = 0.0, collections.Counter()
        for op, c in dyn.items():
            cy = cycles_of(op, probe)
            if cy is None:
                unprobed[op] += c
                cy = 4.0
            cyc += c * cy
        avg = cyc / total
        out["dynamic"] = {"workload": args.workload, "passes": passes, "predicted_valu_wave_instructions": total, "per_phase": per_phase,
                          "histogram": dict(dyn.most_common()), "unprobed_opcodes_priced_at_4_cycles": dict(unprobed.most_common()),
                          "unprobed_share": round(sum(unprobed.values()) / total, 4)}
        out["mix"] = {"average_cycles_per_valu_instruction": round(avg, 4),
                      "mix_ceiling_G_wave_instructions_s": round(1024 * CLOCK_GHZ / avg, 1),
                      "mix_ceiling_frac_of_peak": round(2.0 / avg, 4),
                      "peak_G_wave_instructions_s": 1024 * CLOCK_GHZ / 2.0,
                      "cycles_source": "profiles/r02_valu_probe.txt (tools/archive/valu_probe.hip: ns per wave-instruction per SIMD at 6 waves / SIMD) x 2.4 GHz"}
        t3 = probe3_tables()
        if t3:
            rep = {}
            for which, name in ((0, "static_operands"), (1, "varied_operands")):
                a3 = sum(c * cycles3(op, t3, which) for op, c in dyn.items()) / total
                rep[name] = {"average_cycles_per_valu_instruction": round(a3, 4), "mix_ceiling_G_wave_instructions_s": round(1024 * CLOCK_GHZ / a3, 1),
                             "mix_ceiling_frac_of_peak": round(2.0 / a3, 4)}
            rep["source"] = ("profiles/r03_valu_probe3.txt (tools/archive/gen_valu_probe.py -> tools/archive/valu_probe3.hip): the same histogram priced with a probe that pins "
                             "physical registers and separates the destination from the sources, once on operands that never change and once on "
                             "lane-varying ones; the full-rate opcodes (v_mul / v_add / v_sub_f32, v_add_u32, v_and_b32 ...) cost 2.5-2.8 cycles in the "
                             "first setting and 3.8-4.2 in the second in their VOP2 encoding, the quarter-rate ones 4.5-4.7 in both")
            out["mix"]["repriced_with_pinned_register_probe"] = rep
        if os.path.exists(args.counters) and args.workload == "c3":
            measured = json.load(open(args.counters))["counters_mean_per_launch"].get("SQ_INSTS_VALU")
            if measured:
                out["dynamic"]["measured_SQ_INSTS_VALU"] = measured
                out["dynamic"]["predicted_over_measured"] = round(total / measured, 4)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
