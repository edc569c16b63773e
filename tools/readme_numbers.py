#!/usr/bin/env python3
"""The measured figures README.md quotes, generated from a bench record so that they cannot drift or be rounded up.

    python3 tools/readme_numbers.py            # print the block
    python3 tools/readme_numbers.py --write    # replace the block between the markers in README.md

Source: the record of the highest round among the driver's BENCH_rNN.json (its `parsed` line) and the builder's profiles/rNN_bench.json
(`python3 bench.py` on one MI355X with the round's counter files in place); the driver's wins a tie.  Figures are truncated, never rounded up.
tests/test_readme_numbers.py checks that README.md holds exactly the block of the source it names and that the source is of the newest round."""
import glob
import json
import math
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- numbers:begin (tools/readme_numbers.py) -->", "<!-- numbers:end -->"


def sources():
    out = []
    for f in glob.glob(os.path.join(ROOT, "BENCH_r*.json")):
        m = re.search(r"BENCH_r(\d+)\.json$", f)
        try:
            d = json.load(open(f))
        except ValueError:
            continue
        p = d.get("parsed") or {}
        if m and p.get("value"):
            out.append((int(m.group(1)), 1, os.path.relpath(f, ROOT), p))
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")):
        m = re.search(r"r(\d+)_bench\.json$", f)
        d = json.load(open(f))
        if m and d.get("value"):
            out.append((int(m.group(1)), 0, os.path.relpath(f, ROOT), d))
    return sorted(out)


def g(v, digits=2):
    """Mrays/s -> Grays/s, truncated"""
    f = 10 ** digits
    return f"{math.floor(v / 1000.0 * f) / f:.{digits}f}"


def block(rel, b):
    e = b.get("extras") or {}
    r = b.get("roofline") or {}
    cb = b.get("cpu_baseline") or {}
    L = [BEGIN, f"Measured on one MI355X; every figure below is read from `{rel}` (truncated, not rounded):", ""]
    L.append("| workload | Grays/s |")
    L.append("|---|---|")
    if "repeated_value" in b:   # round 6 on: `value` is the first launch of a batch
        L.append(f"| **C3** — 1 048 576-triangle instanced TLAS, 4 194 304 primary rays, `closest_hit`, every launch a batch never traced before, rays read from HBM "
                 f"(`value`; {b['ms_per_step']:.4f} ms per step) | **{g(b['value'])}** |")
        if b.get("same_buffer_natural_order_value"):
            L.append(f"| ... one cache-warm ray buffer, natural claim order (`same_buffer_natural_order_value`; rounds 4-5's `first_touch_value`) | {g(b['same_buffer_natural_order_value'])} |")
        if b.get("repeated_value"):
            L.append(f"| ... one buffer replayed with the claim order learned from its earlier launches (`repeated_value`; rounds 1-5's `value`) | {g(b['repeated_value'])} |")
    else:
        L.append(f"| **C3** — 1 048 576-triangle instanced TLAS, 4 194 304 primary rays, `closest_hit`, the batch repeated (`value`; {b['ms_per_step']:.4f} ms per step) | **{g(b['value'])}** |")
        if b.get("first_touch_value"):
            L.append(f"| the same batch traced for the first time (natural claim order; `first_touch_value`) | {g(b['first_touch_value'])} |")
    if b.get("moving_camera_value"):
        L.append(f"| a camera that moves every frame (`moving_camera_value`) | {g(b['moving_camera_value'])} |")
    rows = [("c3_any_hit_shadow_mrays_s", "C3 shadow rays, `any_hit`, 2.08 M rays"), ("c3_any_hit_shadow_first_launch_mrays_s", "... first launch"),
            ("c4_incoherent_16M_closest_mrays_s", "C4 — 16 777 216 incoherent bounce rays"), ("c3_1Mi_primary_closest_mrays_s", "C3, 1 Mi primary rays"),
            ("c2_100k_blas_1M_coherent_closest_mrays_s", "C2 — 100 000-triangle BLAS, 1 M coherent rays"), ("c2_100k_blas_1M_coherent_closest_first_launch_mrays_s", "... first launch"),
            ("c2_100k_blas_1M_coherent_closest_4_in_flight_mrays_s", "... four launches in flight")]
    for k, name in rows:
        if isinstance(e.get(k), (int, float)):
            L.append(f"| {name} | {g(e[k])} |")
    b4 = e.get("c2_4_independent_batches_one_call") or {}
    if b4.get("mrays_s"):
        L.append(f"| four independent C2-sized batches in ONE call (`rc_trace_closest_device_batches`; one after the other: {g(b4['one_after_the_other_mrays_s'])}) | {g(b4['mrays_s'])} |")
    big = e.get("c3_blas_more_instances_closest") or {}
    for k in sorted(big, key=int):
        L.append(f"| {k} instances of the C3 BLAS ({big[k]['triangles']:,} triangles) | {g(big[k]['mrays_s'])} |".replace(",", " "))
    L.append("")
    ref = e.get("random_geometry_1M_rays_closest") or {}
    if ref:
        L.append("The reference's own traversal benchmark (random geometry in one BLAS, 1 M rays, `closest_hit`; `benchmarks/implicitbvh_comparison.md:37-39`, RX 7900 XTX): "
                 + "; ".join(f"{int(k):,} triangles {v['ms_per_1M_rays']:.3f} ms ({v['reference_rx7900xtx_ms']} ms there)".replace(",", " ") for k, v in sorted(ref.items(), key=lambda kv: int(kv[0]))) + ".")
    bl = e.get("blas_build_device") or {}
    if bl:
        L.append("BLAS build, triangles already in HBM: " + "; ".join(f"{int(k):,} triangles {v['ms']:.3f} ms ({v['reference_rx7900xtx_ms']} ms there)".replace(",", " ") for k, v in sorted(bl.items(), key=lambda kv: int(kv[0]))) + ".")
    vf = e.get("view_factors_c5") or {}
    if vf:
        parts = []
        if vf.get("totals_ms"):
            parts.append(f"per-triangle totals {vf['totals_ms']:.1f} ms")
        if (vf.get("rows_sharded") or {}).get("seconds"):
            parts.append(f"device-resident matrix {vf['rows_sharded']['seconds'] * 1e3:.1f} ms")
        hm = (vf.get("host_matrix_e2e") or {})
        if parts:
            L.append(f"`view_factors` at C5 ({vf.get('n_prims')} triangles, {vf.get('n_rays', 0) / 1e6:.1f} M rays): " + ", ".join(parts) + ".")
    for key, what in (("hbm_regime_4M_tris_4M_incoherent_rays", "4 M-triangle BLAS (512 MB of nodes)"), ("hbm_regime_16M_tris_4M_incoherent_rays", "16 M-triangle BLAS (2 GB of nodes, past the 256 MiB Infinity Cache)")):
        h = e.get(key) or {}
        rf = h.get("roofline") or {}
        if rf.get("bound") == "texture-path":
            L.append(f"Memory-bound regime, {what}, 4 M incoherent rays: {g(h['mrays_s'])} Grays/s; bound by the texture data path (TD {rf['td_busy_frac']:.2f} busy); "
                     f"{rf['hbm_fabric_GBs']:.0f} GB/s requested from the fabric = {rf['hbm_fabric_frac']:.2f} of the 8 TB/s HBM peak (Infinity-Cache hits included), "
                     f"{rf['frac_of_achievable_random']:.2f} x the 2.5 TB/s of the round-2 probe of DEPENDENT random 64-byte gathers (which was therefore not the machine's limit for this pattern).")
        elif rf:
            L.append(f"Where HBM binds ({what}, 4 M incoherent rays): {g(h['mrays_s'])} Grays/s at {rf['achieved']:.0f} GB/s of physical HBM traffic "
                     f"= {rf['frac']:.2f} of the 8 TB/s peak, {rf['frac_of_achievable']:.2f} of the achievable 6.3.")
    if r.get("frac"):
        L.append(f"Headline kernel: VALU issue {r['frac']:.2f} of the guide's 2-cycle peak, {r.get('lane_utilisation', 0):.2f} of the issued lanes carry a ray, "
                 f"waves waiting {r.get('waiting_frac_of_wave_cycles') or 0:.2f} of their cycles, texture-data path {r.get('td_busy_frac') or 0:.2f} busy, physical HBM {r.get('hbm_physical_frac') or 0:.2f} of peak.")
    if cb.get("value"):
        L.append(f"CPU baseline (the oracle, {cb.get('kind')}): {cb['value']:.1f} {cb['unit']} on {cb['cores']} threads.")
    L.append(END)
    return "\n".join(L)


def current():
    src = sources()
    if not src:
        raise SystemExit("no bench record found")
    rnd, _, rel, b = src[-1]
    return rnd, rel, block(rel, b)


def main():
    rnd, rel, text = current()
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "README.md")
        s = open(p).read()
        if BEGIN in s and END in s:
            s = s[:s.index(BEGIN)] + text + s[s.index(END) + len(END):]
        else:
            s = s.rstrip("\n") + "\n\n" + text + "\n"
        open(p, "w").write(s)
    else:
        print(text)


if __name__ == "__main__":
    main()
