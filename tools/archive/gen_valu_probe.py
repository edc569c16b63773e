#!/usr/bin/env python3
"""Writes tools/valu_probe3.hip: the per-opcode VALU issue cost table the roofline's `mix_ceiling` is priced with.

Why a third probe: the first one (tools/valu_probe.hip, profiles/r02_valu_probe.txt) let the compiler pick registers and ran every opcode on
whatever values its chain happened to produce (products decaying to zero, sums running to infinity...).  tools/valu_probe2.hip showed that
this matters more than the opcode's encoding: the SAME v_mul_f32 costs 2.8 cycles (at the nominal 2.4 GHz) on operands that never change
and 3.1-3.6 on operands with busy, lane-varying mantissas -- the chip runs all 1024 SIMDs flat out here and manages its clock by
power, so time per instruction depends on how many bits toggle.  This probe therefore measures every opcode of the traversal kernel's
mix in ONE setting: physical registers pinned (sources and destinations in known VGPR banks, destination separate from the sources as in
compiled code), 8 independent instructions per loop trip, 6 waves per SIMD (the kernel's occupancy), once with every register = 1.0
("static": the optimistic bound) and once with lane- and register-varying operands ("varied": what a traversal's operands look like).
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def A(k): return f"v{24 + k}"
def B(k): return f"v{40 + k}"
def Cc(k): return f"v{48 + k}"
def D(k): return f"v{56 + k}"
def PA(k): return f"v[{24 + 2 * k}:{25 + 2 * k}]"
def PB(k): return f"v[{40 + 2 * k}:{41 + 2 * k}]"
def PD(k): return f"v[{56 + 2 * k}:{57 + 2 * k}]"
def S(k): return f"s[{52 + 2 * k}:{53 + 2 * k}]"
def S1(k): return f"s{52 + k}"


FORMS = [
    ("v_mul_f32_e32", lambda k: f"v_mul_f32_e32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_add_f32_e32", lambda k: f"v_add_f32_e32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_sub_f32_e32", lambda k: f"v_sub_f32_e32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_mul_f32_e64", lambda k: f"v_mul_f32_e64 {D(k)}, {A(k)}, -{B(k)}"),
    ("v_fma_f32", lambda k: f"v_fma_f32 {D(k)}, {A(k)}, {B(k)}, {Cc(k)}"),
    ("v_fmac_f32_e32", lambda k: f"v_fmac_f32_e32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_pk_mul_f32", lambda k: f"v_pk_mul_f32 {PD(k)}, {PA(k)}, {PB(k)}"),
    ("v_pk_add_f32", lambda k: f"v_pk_add_f32 {PD(k)}, {PA(k)}, {PB(k)}"),
    ("v_pk_fma_f32", lambda k: f"v_pk_fma_f32 {PD(k)}, {PA(k)}, {PB(k)}, {PB((k + 1) % 8)}"),
    ("v_maximum3_f32", lambda k: f"v_maximum3_f32 {D(k)}, {A(k)}, {B(k)}, {Cc(k)}"),
    ("v_minimum3_f32", lambda k: f"v_minimum3_f32 {D(k)}, {A(k)}, {B(k)}, {Cc(k)}"),
    ("v_max_f32_e32", lambda k: f"v_max_f32_e32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_max3_f32", lambda k: f"v_max3_f32 {D(k)}, {A(k)}, {B(k)}, {Cc(k)}"),
    ("v_cmp_f32_e64", lambda k: f"v_cmp_lt_f32_e64 {S(k)}, {A(k)}, {B(k)}"),
    ("v_cmp_f32_e32", lambda k: f"v_cmp_lt_f32_e32 vcc, {A(k)}, {B(k)}"),
    ("v_cmp_u32_e64", lambda k: f"v_cmp_lt_u32_e64 {S(k)}, {A(k)}, {B(k)}"),
    ("v_cmp_u32_e32", lambda k: f"v_cmp_lt_u32_e32 vcc, {A(k)}, {B(k)}"),
    ("v_cndmask_b32_e64", lambda k: f"v_cndmask_b32_e64 {D(k)}, {A(k)}, {B(k)}, s[68:69]"),
    ("v_cndmask_b32_e32", lambda k: f"v_cndmask_b32_e32 {D(k)}, {A(k)}, {B(k)}, vcc"),
    ("v_mov_b32_e32", lambda k: f"v_mov_b32_e32 {D(k)}, {A(k)}"),
    ("v_mov_b64", lambda k: f"v_mov_b64 {PD(k)}, {PA(k)}"),
    ("v_add_u32_e32", lambda k: f"v_add_u32_e32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_sub_u32_e32", lambda k: f"v_sub_u32_e32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_and_b32_e32", lambda k: f"v_and_b32_e32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_lshrrev_b32_e32", lambda k: f"v_lshrrev_b32_e32 {D(k)}, 3, {A(k)}"),
    ("v_lshlrev_b32_e32", lambda k: f"v_lshlrev_b32_e32 {D(k)}, 3, {A(k)}"),
    ("v_lshl_add_u32", lambda k: f"v_lshl_add_u32 {D(k)}, {A(k)}, 3, {B(k)}"),
    ("v_add_lshl_u32", lambda k: f"v_add_lshl_u32 {D(k)}, {A(k)}, {B(k)}, 3"),
    ("v_add3_u32", lambda k: f"v_add3_u32 {D(k)}, {A(k)}, {B(k)}, {Cc(k)}"),
    ("v_bfi_b32", lambda k: f"v_bfi_b32 {D(k)}, {A(k)}, {B(k)}, {Cc(k)}"),
    ("v_lshl_add_u64", lambda k: f"v_lshl_add_u64 {PD(k)}, {PA(k)}, 2, {PB(k)}"),
    ("v_lshlrev_b64", lambda k: f"v_lshlrev_b64 {PD(k)}, 3, {PA(k)}"),
    ("v_lshrrev_b64", lambda k: f"v_lshrrev_b64 {PD(k)}, 3, {PA(k)}"),
    ("v_mad_u64_u32", lambda k: f"v_mad_u64_u32 {PD(k)}, {S(k)}, {A(k)}, {B(k)}, {PB(k)}"),
    ("v_mul_hi_u32", lambda k: f"v_mul_hi_u32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_mul_lo_u32", lambda k: f"v_mul_lo_u32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_rcp_f32_e32", lambda k: f"v_rcp_f32_e32 {D(k)}, {A(k)}"),
    ("v_div_scale_f32", lambda k: f"v_div_scale_f32 {D(k)}, vcc, {A(k)}, {B(k)}, {A(k)}"),
    ("v_div_fmas_f32", lambda k: f"v_div_fmas_f32 {D(k)}, {A(k)}, {B(k)}, {Cc(k)}"),
    ("v_div_fixup_f32", lambda k: f"v_div_fixup_f32 {D(k)}, {A(k)}, {B(k)}, {Cc(k)}"),
    ("v_readfirstlane_b32", lambda k: f"v_readfirstlane_b32 {S1(k)}, {A(k)}"),
    ("v_mbcnt_lo_u32_b32", lambda k: f"v_mbcnt_lo_u32_b32 {D(k)}, -1, {A(k)}"),
    ("v_mbcnt_hi_u32_b32", lambda k: f"v_mbcnt_hi_u32_b32 {D(k)}, -1, {A(k)}"),
    # the VOP3 encodings of the full-rate opcodes, and repeats of the VOP2 ones (run order / clock history check)
    ("v_mul_f32_e64 plain", lambda k: f"v_mul_f32_e64 {D(k)}, {A(k)}, {B(k)}"),
    ("v_mul_f32_e32 again", lambda k: f"v_mul_f32_e32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_add_f32_e64", lambda k: f"v_add_f32_e64 {D(k)}, {A(k)}, {B(k)}"),
    ("v_add_f32_e32 again", lambda k: f"v_add_f32_e32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_sub_f32_e64", lambda k: f"v_sub_f32_e64 {D(k)}, {A(k)}, {B(k)}"),
    ("v_fmac_f32_e64", lambda k: f"v_fmac_f32_e64 {D(k)}, {A(k)}, {B(k)}"),
    ("v_mov_b32_e64", lambda k: f"v_mov_b32_e64 {D(k)}, {A(k)}"),
    ("v_add_u32_e64", lambda k: f"v_add_u32_e64 {D(k)}, {A(k)}, {B(k)}"),
    ("v_and_b32_e64", lambda k: f"v_and_b32_e64 {D(k)}, {A(k)}, {B(k)}"),
    ("v_lshrrev_b32_e64", lambda k: f"v_lshrrev_b32_e64 {D(k)}, 3, {A(k)}"),
    ("v_mul_f32_e32 3rd", lambda k: f"v_mul_f32_e32 {D(k)}, {A(k)}, {B(k)}"),
    ("v_cndmask_b32_e64 vcc", lambda k: f"v_cndmask_b32_e64 {D(k)}, {A(k)}, {B(k)}, vcc"),
    ("v_cndmask_b32_e32 const", lambda k: f"v_cndmask_b32_e32 {D(k)}, 0, {B(k)}, vcc"),
    ("v_mul_f32_e32 d==s0", lambda k: f"v_mul_f32_e32 {A(k)}, {A(k)}, {B(k)}"),
    ("v_mul_f32_e64 d==s0", lambda k: f"v_mul_f32_e64 {A(k)}, {A(k)}, {B(k)}"),
]

HEADER = r'''// GENERATED by tools/gen_valu_probe.py -- edit the generator, not this file.
// Per-opcode VALU issue cost on gfx950 at the traversal kernel's occupancy (6 waves per SIMD), physical registers pinned, destination
// separate from the sources, 8 independent instructions per loop trip; "static" = every register 1.0, "varied" = lane- and
// register-varying operands (the chip manages its clock by power: busy operands cost 10-25 % more time per instruction).
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe3.hip -o tools/valu_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
'''


def main():
    vregs = ", ".join(f'"v{r}"' for r in range(24, 72))
    sregs = ", ".join(f'"s{r}"' for r in range(52, 70))
    out = [HEADER, f"#define CLOBBERS {vregs}, {sregs}, \"vcc\"\n"]
    out.append("template <int V>\n__global__ __launch_bounds__(256, 6) void k_probe(float* out, int iters, int varied) {\n")
    out.append("    const unsigned t = threadIdx.x;\n")
    # initial values: 1.0 everywhere, or 1 + a lane- and register-dependent fraction (all finite, no denormals; the integer forms use the same bits)
    out.append("    asm volatile(\"s_mov_b32 s68, 0x5555aaaa\\n s_mov_b32 s69, 0xaaaa5555\\n s_mov_b32 vcc_lo, 0x3333cccc\\n s_mov_b32 vcc_hi, 0xcccc3333\" : : : CLOBBERS);\n")
    for r in range(24, 72):
        out.append(f"    {{ const float x = varied ? 1.0f + (float)((t * 37u + {r * 11}u) % 251u) * 0x1p-9f + (float)((t + {r}u) % 7u) * 0x1p-22f : 1.0f; asm volatile(\"v_mov_b32 v{r}, %0\" : : \"v\"(x) : CLOBBERS); }}\n")
    out.append("    for (int i = 0; i < iters; ++i) {\n")
    for v, (name, f) in enumerate(FORMS):
        body = "\\n ".join(f(k) for k in range(8))
        out.append(f"        if (V == {v}) asm volatile(\"{body}\" : : : CLOBBERS);\n")
    out.append("    }\n    float acc;\n    asm volatile(\"v_add_f32 %0, v56, v57\\n v_add_f32 %0, %0, v63\\n v_add_f32 %0, %0, v71\" : \"=v\"(acc) : : CLOBBERS);\n")
    out.append("    out[blockIdx.x * 256 + threadIdx.x] = acc;\n}\n\n")
    out.append(r'''template <int V>
static int run(const char* name, float* d_out, int n_cus) {
    const int iters = 20000, blocks = n_cus * 6;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double ns[2];
    for (int varied = 0; varied < 2; ++varied) {
        hipLaunchKernelGGL((k_probe<V>), dim3(blocks), dim3(256), 0, 0, d_out, 200, varied);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_probe<V>), dim3(blocks), dim3(256), 0, 0, d_out, iters, varied);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        ns[varied] = best * 1e6 / (6.0 * iters * 8);
    }
    printf("%-22s static %.3f ns = %.2f cycles   varied %.3f ns = %.2f cycles   (per wave-instruction per SIMD, cycles at 2.4 GHz)\n", name, ns[0], ns[0] * 2.4, ns[1], ns[1] * 2.4);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n = prop.multiProcessorCount;
    float* d;
    CK(hipMalloc(&d, sizeof(float) * n * 6 * 256));
    printf("device %s, %d CUs; 6 waves / SIMD, 8 independent instructions per trip, destination separate from the sources\n", prop.name, n);
''')
    for v, (name, f) in enumerate(FORMS):
        out.append(f"    if (run<{v}>(\"{name}\", d, n)) return 1;\n")
    out.append("    return 0;\n}\n")
    path = os.path.join(ROOT, "tools", "valu_probe3.hip")
    open(path, "w").write("".join(out))
    print("wrote", path, len(FORMS), "forms")


if __name__ == "__main__":
    main()
