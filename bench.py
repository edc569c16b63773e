#!/usr/bin/env python3
"""bench.py -- Mrays/s of closest_hit on the 1M-triangle instanced TLAS (BASELINE.json config C3).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one closest_hit pass over one batch of 4 194 304 synthetic primary rays (2048 x 2048 pinhole) against
256 transformed instances of one 4 096-triangle BLAS (1 048 576 triangles), rays and scene already resident in
HBM.  One process per GPU; rays are independent, so ranks are replicas of the scene tracing their own batch
(no data-path collective; "scaling": "weak").  Rank 0 prints ONE JSON line.  Started as a plain `python bench.py
--gpus N` (no WORLD_SIZE in the environment) with N > 1, the script launches its own N rank processes -- before anything
touches a GPU -- and relays rank 0's line.

roofline: the dominant kernel (k_trace_phased_lds) is bound by VALU ISSUE, not by memory -- the scene is LDS / L1 / L2 resident
and physical HBM traffic is the ray-in / hit-out stream (6 % of the HBM peak).  `achieved` = VALU wave-instructions per launch
(SQ_INSTS_VALU from the counter file named in `roofline.sources`, collected with rocprofv3 --pmc on this kernel and workload; the file
carries a fingerprint of the kernel sources and is ignored when it does not match) / the average launch duration measured live with HIP
events on the launch stream; `peak` = the guide's issue peak, 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction =
1228.8 G/s.  `mix_ceiling` = what this kernel's opcode mix could reach at the cycles measured per opcode (tools/isa_mix.py);
`lane_utilisation` = the share of issued lane slots that carry a ray; `lane_throughput_frac` = frac x lane_utilisation.
The figure SURVEY.md section 8d prescribes -- algorithmic bytes of the REFERENCE algorithm (32 + 32 + 60 x BVHNode2 fetches + 140 x
TLAS-leaf entries per ray, counted by the instrumented CPU oracle in the cpu_baseline leg) against the 8 TB/s HBM peak -- is kept
as `algorithmic_vs_hbm`: it exceeds 1 because those bytes come from LDS and caches, which is why it is not `frac`.
cpu_baseline: the reference algorithm's C restatement (oracle/), all host cores and one thread, same rays.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
HBM_ACHIEVABLE_GBS = 6300.0  # same guide: ~6.3 TB/s achievable
HBM_RANDOM64_GBS = 2500.0  # dependent random 64-byte gathers: profiles/r02_fetch_calibration.txt (k_gather64_deep, 39 G requests/s)
# The guide's execution model: a wave64 VALU instruction issues over 2 cycles on the SIMD-32 (157.3 TF f32 vector peak):
# 256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1228.8 G wave-instructions / s.  (Round 2 divided by 4 cycles -- the measured cost of v_mul / v_add /
# v_mov -- which the builder's own probe contradicts for v_fma_f32; the per-opcode measurements now enter through `mix_ceiling`.)
VALU_PEAK_GINST_S = 256 * 4 * 2.4 / 2.0
COUNTER_FILE = os.path.join("profiles", "r06_pmc_c3.json")          # per-launch PMC counters of the bench kernel (tools/capture_profiles.sh)
WORKLOADS_FILE = os.path.join("profiles", "r06_pmc_workloads_kernel5.json")  # the same counters for the extras' workloads: C2, shadow rays, C4, random geometry, C3 1 Mi rays, the HBM-bound regime (tools/pmc_workloads.sh)
MIX_FILE = os.path.join("profiles", "r06_isa_mix_kernel5.json")     # dynamic opcode histogram of the bench kernel x measured cycles per opcode (tools/isa_mix.py)
COUNTS_FILE = os.path.join("profiles", "c3_reference_counts.json")  # reference-algorithm fetch counts per ray for this workload (written by the N=1 run)
# Fallback when the counts file is missing (same numbers, measured by the oracle in round 1)
C3_NODE_FETCHES_PER_RAY = 33.006
C3_INST_ENTRIES_PER_RAY = 1.922


def kernel_fingerprint():
    """sha256 over what the dominant kernel is compiled from (the traversal core, the device records, the kernel file, the compiler
    flags).  Stored in the counter file by tools/capture_profiles.sh, recomputed by every bench run."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "raycore.jl_amd", "csrc")
    for name in ("rc_traverse_core.h", "rc_device.h", "rc_traverse.hip"):
        h.update(open(os.path.join(csrc, name), "rb").read())
    for line in open(os.path.join(csrc, "Makefile")):
        if line.startswith("FLAGS") or line.startswith("        -W"):
            h.update(line.encode())
    return {"sha256_of": "raycore.jl_amd/csrc/{rc_traverse_core.h, rc_device.h, rc_traverse.hip} + the Makefile's FLAGS", "sha256": h.hexdigest()}


def make_roofline(launch_ms, n_rays, node_f, inst_f, counts_source, pmc, kernel_option=-1, profiled_config=True, mix=None, fingerprint=None):
    """The `roofline` object of the JSON line from a launch time (ms, measured live), the reference algorithm's fetch counts per ray
    and the per-launch PMC counters of the counter file.  Pure arithmetic, so a reader can recompute every number from profiles/."""
    bytes_per_ray = 32 + 32 + 60.0 * node_f + 140.0 * inst_f
    alg_gbs = bytes_per_ray * n_rays / (launch_ms * 1e-3) / 1e9
    c = pmc.get("counters_mean_per_launch", {})
    valu = c.get("SQ_INSTS_VALU")
    traffic = (pmc.get("hbm") or {}).get("c3_closest_bytes_per_launch")
    # the kernel's name as rocprofv3 reported it when the counters were taken (a literal here had drifted from the template's arity)
    kname = (pmc.get("kernel") or {}).get("Kernel_Name") if kernel_option in (-1, 5) else None
    kname = kname or {-1: "k_trace_phased_lds (auto)", 5: "k_trace_phased_lds", 3: "k_trace_phased"}.get(kernel_option, f"kernel option {kernel_option}")
    stale = None
    if fingerprint is not None and (pmc.get("fingerprint") or {}).get("sha256") != fingerprint.get("sha256"):
        stale = f"stale counters: {COUNTER_FILE} was captured from other kernel sources (fingerprint mismatch); re-run tools/capture_profiles.sh"
    if valu and profiled_config and not stale:
        achieved = valu / (launch_ms * 1e-3) / 1e9
        frac = achieved / VALU_PEAK_GINST_S
        lane_util = c["SQ_THREAD_CYCLES_VALU"] / (valu * 64.0) if c.get("SQ_THREAD_CYCLES_VALU") else None
        roofline = {"bound": "valu-issue", "achieved": round(achieved, 1), "peak": round(VALU_PEAK_GINST_S, 1), "unit": "G wave-instructions/s",
                    "frac": round(frac, 4), "traffic": traffic, "kernel": kname, "avg_launch_ms": round(launch_ms, 4),
                    "valu_wave_instructions_per_launch": valu,
                    "lane_utilisation": round(lane_util, 4) if lane_util else None,
                    # of the chip's lane-instruction throughput (64 lanes x the issue peak), the share that did work for a ray
                    "lane_throughput_frac": round(frac * lane_util, 4) if lane_util else None,
                    "hbm_physical_frac": round(traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                    "vmem_wave_instructions_per_launch": c.get("SQ_INSTS_VMEM_RD"),
                    # round 5 (profiles/r05_bound_probe.txt, r05_td_model.txt): what the waves wait for.  TD is busy per fetch INSTRUCTION whatever the exec mask.
                    "waiting_frac_of_wave_cycles": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAIT_ANY") and c.get("SQ_WAVE_CYCLES") else None,
                    "td_busy_frac": round(c["TD_TD_BUSY_sum"] / c["GRBM_GUI_ACTIVE"] / 32.0, 4) if c.get("TD_TD_BUSY_sum") and c.get("GRBM_GUI_ACTIVE") else None,
                    "ta_busy_frac": round(c["TA_TA_BUSY_sum"] / c["GRBM_GUI_ACTIVE"] / 32.0, 4) if c.get("TA_TA_BUSY_sum") and c.get("GRBM_GUI_ACTIVE") else None,
                    "what_bounds_it": "the length of each ray's chain of dependent passes (fetch through the texture path -> ~8 dependent VALU levels -> stack access) at a fixed number of rays "
                                      "in flight: serial work added to a pass costs 2.3-2.5x its length, VALU work removed beside the chain buys nothing (12 of the 20 min / max of every C2 pass: "
                                      "SQ_INSTS_VALU -8.7 %, launch time unchanged).  Right behind it: the texture-data path, busy `td_busy_frac` of the launch with the four fetch instructions of every node visit (~22 cycles each "
                                      "whatever the exec mask) -- a shorter chain could buy at most 1 / td_busy_frac.  `frac` stays the VALU-issue figure of earlier rounds for continuity; it is not the binding roof.",
                    "sources": {"valu_wave_instructions_per_launch, lane_utilisation, traffic": COUNTER_FILE + " (rocprofv3 --pmc passes over this bench command, tools/capture_profiles.sh; its fingerprint matches the kernel sources of this run; FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md: the traffic is the coalesced ray / hit stream)",
                                "avg_launch_ms": "the HIP events every timed launch carries on its own kernel dispatch (start / end of the kernel on the launch stream), read back after the timed region: rc_recent_kernel_ms, this run",
                                "peak": "MI355X_MICROARCH.md execution model: a wave64 VALU instruction issues over 2 cycles on the SIMD-32: 256 CUs x 4 SIMDs x 2.4 GHz / 2",
                                "node / instance counts": counts_source}}
        m = (mix or {}).get("mix")
        if m:
            ceiling = m["mix_ceiling_G_wave_instructions_s"]
            roofline["mix_ceiling"] = {"G_wave_instructions_s": ceiling, "frac_of_peak": round(ceiling / VALU_PEAK_GINST_S, 4),
                                       "average_cycles_per_valu_instruction": m["average_cycles_per_valu_instruction"],
                                       "achieved_over_mix_ceiling": round(achieved / ceiling, 4),
                                       "note": "the issue rate this kernel's dynamic opcode mix could reach on 1024 SIMDs at the cycles measured per opcode (v_pk_*_f32 4.9, "
                                               "v_minimum3 / v_maximum3 4.5, compares 4.9, moves 3.8 ...); none of them issues in the guide's 2 cycles",
                                       "source": MIX_FILE + " (tools/isa_mix.py: per-phase opcode histogram of the kernel's ISA x the STATS kernel's pass counts x profiles/r02_valu_probe.txt)"}
            rep = m.get("repriced_with_pinned_register_probe")
            if rep:  # the same histogram priced with the second probe's two tables: how far the per-opcode costs themselves are uncertain
                lo, hi = rep["varied_operands"]["mix_ceiling_G_wave_instructions_s"], rep["static_operands"]["mix_ceiling_G_wave_instructions_s"]
                roofline["mix_ceiling"]["range_G_wave_instructions_s"] = [lo, hi]
                roofline["mix_ceiling"]["achieved_over_mix_ceiling_range"] = [round(achieved / hi, 4), round(achieved / lo, 4)]
                roofline["mix_ceiling"]["range_note"] = ("per-opcode issue costs depend on the operands' bit activity and on the encoding (full-rate VOP2 opcodes: 2.5-2.8 cycles on "
                                                         "static operands, 3.8-4.2 on lane-varying ones); the range prices the histogram with both tables of profiles/r03_valu_probe3.txt")
    else:  # no usable counter file for this configuration: only the section-8d figure can be given, and it is not a utilisation
        roofline = {"bound": "valu-issue", "achieved": None, "peak": round(VALU_PEAK_GINST_S, 1), "unit": "G wave-instructions/s", "frac": None, "traffic": traffic,
                    "kernel": kname, "avg_launch_ms": round(launch_ms, 4),
                    "sources": {"note": stale or f"{COUNTER_FILE} missing or --res differs from the profiled 2048"}}
    roofline["algorithmic_vs_hbm"] = {"achieved_GBs": round(alg_gbs, 1), "peak_GBs": HBM_PEAK_GBS, "ratio": round(alg_gbs / HBM_PEAK_GBS, 4),
                                      "algorithmic_bytes_per_ray": round(bytes_per_ray, 1), "node_fetches_per_ray": round(node_f, 3), "instance_entries_per_ray": round(inst_f, 3),
                                      "note": "SURVEY section 8d's figure; > 1 because the reference algorithm's node / instance bytes are served from LDS, L1 and L2, not HBM"}
    return roofline


def make_hbm_regime(rate_mrays_s, node_fetches_per_ray, n_rays, launch_ms, entry, fingerprint_ok, source=WORKLOADS_FILE, key="hbm", tree_bytes=None):
    """The memory-bound extras: rate and node fetches measured live; bytes per launch from the workload's FETCH_SIZE /
    WRITE_SIZE passes in the per-workload counter file and TD / TA busy from the same file -- used only when the file's fingerprint matches the
    kernel sources of this run.  What bounds these launches is the texture data path (TD ~0.98 busy), not HBM: `roofline.bound` says so, the
    primary fraction is TD busy, and the memory-side figure is reported as `hbm_fabric_frac` -- FETCH_SIZE counts every request L2 sends to the
    fabric, Infinity-Cache (MALL, 256 MiB) hits included, so it is an UPPER bound on DRAM traffic for a tree that partly fits the MALL (the 512 MB
    tree) and close to DRAM traffic for one that does not (the 2 GB tree)."""
    alg = (64 + 60.0 * node_fetches_per_ray + 140.0) * n_rays
    secs = n_rays / (rate_mrays_s * 1e6)
    out = {"mrays_s": rate_mrays_s, "node_fetches_per_ray": round(node_fetches_per_ray, 2), "algorithmic_GBs": round(alg / secs / 1e9, 1), "tree_bytes": tree_bytes}
    h = (entry or {}).get("hbm") if fingerprint_ok else None
    if not h:
        out.update({"hbm_physical_GBs": None, "hbm_physical_frac": None, "fetch_amplification": None, "roofline": None,
                    "note": f"{source} missing, without the '{key}' workload, or captured from other kernel sources (fingerprint): run tools/pmc_workloads.sh"})
        return out
    # FETCH_SIZE counts 64 bytes per request: x1 for this workload's random 64-byte node gathers (profiles/r02_fetch_calibration.txt; the
    # guide's x2 applies to wide coalesced streams -- here only the 268 MB of rays and hits, which x1 under-counts by at most 134 MB)
    phys = h["read_bytes_x1"] + h["write_bytes"]
    gbs = phys / secs / 1e9
    c = (entry or {}).get("counters_mean_per_launch") or {}
    td = c["TD_TD_BUSY_sum"] / c["GRBM_GUI_ACTIVE"] / 32.0 if c.get("TD_TD_BUSY_sum") and c.get("GRBM_GUI_ACTIVE") else None   # (the _sum covers the 32 TDs of one XCD)
    ta = c["TA_TA_BUSY_sum"] / c["GRBM_GUI_ACTIVE"] / 32.0 if c.get("TA_TA_BUSY_sum") and c.get("GRBM_GUI_ACTIVE") else None
    valu = c.get("SQ_INSTS_VALU")
    out.update({"hbm_physical_GBs": round(gbs, 1), "hbm_physical_frac": round(gbs / HBM_PEAK_GBS, 4), "fetch_amplification": round(phys / alg, 3),
                "roofline": {"bound": "texture-path", "achieved": round(td, 4) if td else None, "peak": 1.0, "unit": "TD busy (share of the launch)", "frac": round(td, 4) if td else None,
                             "td_busy_frac": round(td, 4) if td else None, "ta_busy_frac": round(ta, 4) if ta else None,
                             "valu_issue_frac": round(valu / secs / 1e9 / VALU_PEAK_GINST_S, 4) if valu else None,
                             "hbm_fabric_GBs": round(gbs, 1), "hbm_peak_GBs": HBM_PEAK_GBS, "hbm_fabric_frac": round(gbs / HBM_PEAK_GBS, 4),
                             "hbm_fabric_note": "FETCH_SIZE x 64 B + WRITE_SIZE: requests L2 sends to the fabric.  Infinity-Cache hits are counted, so this is an upper bound on DRAM bytes "
                                                "(tree_bytes against the 256 MiB MALL says how loose)",
                             "achievable_random_GBs": HBM_RANDOM64_GBS, "frac_of_achievable_random": round(gbs / HBM_RANDOM64_GBS, 4),
                             "achievable_random_note": "the builder's round-2 probe of DEPENDENT random 64-byte gathers (one chain per lane, profiles/r02_fetch_calibration.txt); a fraction above 1 says "
                                                       "that probe was not the machine's limit for this access pattern -- the traversal keeps more independent requests in flight -- not that a roof was broken",
                             "frac_of_achievable_stream": round(gbs / HBM_ACHIEVABLE_GBS, 4), "achievable_stream_GBs": HBM_ACHIEVABLE_GBS, "traffic": phys,
                             "avg_launch_ms": round(secs * 1e3, 4), "launch_ms_hip_events": launch_ms,
                             "kernel": ((entry.get("kernel") or {}).get("Kernel_Name")),
                             "sources": {"traffic, TD / TA busy": source + f" workloads.{key}: rocprofv3 --pmc, one counter set per pass over tools/perf_probe.py --workloads {key} "
                                                                          "(tools/pmc_workloads.sh), mean of the last three launches; fingerprint matches this run's kernel sources",
                                         "launch time, node fetches": "this run (mean of the batch's last 8 launches; the STATS build of kernel 3 for the fetch count)",
                                         "peaks": "MI355X_MICROARCH.md: HBM3E 8 TB/s peak, ~6.3 TB/s achievable by a coalesced stream; profiles/r02_fetch_calibration.txt: 2.5 TB/s "
                                                  "for dependent random 64-byte gathers (39 G requests/s), this builder's own measurement"}}})
    return out


def make_workload_roofline(entry, launch_ms, n_rays, fingerprint_ok, source=WORKLOADS_FILE):
    """The roofline object of one extra workload from its per-launch counters (tools/pmc_workloads.sh) and a launch time measured live:
    VALU issue against the guide's 2-cycle peak, lane utilisation, the share of the resident waves' cycles spent waiting, the TA / TD data
    path's busy share, physical HBM bytes.  None when the counters describe other kernel sources."""
    c = (entry or {}).get("counters_mean_per_launch") or {}
    valu = c.get("SQ_INSTS_VALU")
    if not valu or not fingerprint_ok or not launch_ms:
        return None
    secs = launch_ms * 1e-3
    achieved = valu / secs / 1e9
    lane = c["SQ_THREAD_CYCLES_VALU"] / (valu * 64.0) if c.get("SQ_THREAD_CYCLES_VALU") else None
    out = {"bound": "valu-issue", "achieved": round(achieved, 1), "peak": round(VALU_PEAK_GINST_S, 1), "unit": "G wave-instructions/s", "frac": round(achieved / VALU_PEAK_GINST_S, 4),
           "avg_launch_ms": round(launch_ms, 4), "valu_wave_instructions_per_launch": valu, "valu_wave_instructions_per_ray": round(valu / n_rays, 2),
           "lane_utilisation": round(lane, 4) if lane else None, "lane_throughput_frac": round(achieved / VALU_PEAK_GINST_S * lane, 4) if lane else None,
           "waiting_frac_of_wave_cycles": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAIT_ANY") and c.get("SQ_WAVE_CYCLES") else None,
           "vmem_wave_instructions_per_launch": c.get("SQ_INSTS_VMEM_RD"), "lds_wave_instructions_per_launch": c.get("SQ_INSTS_LDS"),
           "kernel": (entry.get("kernel") or {}).get("Kernel_Name"), "source": source}
    if c.get("TD_TD_BUSY_sum") and c.get("GRBM_GUI_ACTIVE"):
        out["td_busy_frac"] = round(c["TD_TD_BUSY_sum"] / c["GRBM_GUI_ACTIVE"] / 32.0, 4)  # (the _sum covers the 32 TDs of one XCD, as in profiles/r02_pmc_workloads_kernel5.json)
        out["ta_busy_frac"] = round(c["TA_TA_BUSY_sum"] / c["GRBM_GUI_ACTIVE"] / 32.0, 4) if c.get("TA_TA_BUSY_sum") else None
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        out["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    h = entry.get("hbm")
    if h:
        lo, hi = h["read_bytes_x1"] + h["write_bytes"], h["read_bytes_x2"] + h["write_bytes"]
        out["traffic"] = hi
        out["hbm_physical_frac_range"] = [round(lo / secs / 1e9 / HBM_PEAK_GBS, 4), round(hi / secs / 1e9 / HBM_PEAK_GBS, 4)]
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start N rank processes (fresh children: this process never
    touches a GPU) and relay rank 0's output.  Exit code = the worst child's."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode("utf-8", "replace"))
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--res", type=int, default=2048, help="primary rays = res x res")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo only for dry runs)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercises the launcher, the process group and the JSON line (CPU tests)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.dry_run:
        return dry_run(args, rank, world)
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev:
        if args.backend == "nccl":
            raise SystemExit(f"LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible")
        local_rank %= max(n_dev, 1)  # dry run of the multi-rank control flow on fewer GPUs
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("RC_BENCH_FORCE_DIST") == "1"  # the env switch exercises the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        limit = datetime.timedelta(seconds=300)  # a rank that leaves the collective sequence makes the others fail, not hang
        c0 = time.perf_counter()
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"), timeout=limit)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=limit)
        comm_init_ms = (time.perf_counter() - c0) * 1e3   # rendezvous + communicator (a fixed cost, printed next to the timed calls, never inside them)
    else:
        comm_init_ms = 0.0

    import raycore_jl_amd as rc
    sc = rc.scenes
    cfg = sc.config_c3()
    t = rc.TLAS(local_rank)
    for verts, meta in cfg["blas"]:
        t.add_geometry(verts, meta)
    for b, xf, ids in cfg["instances"]:
        t.push_instances(b, xf, ids)
    t.sync()
    n_tris = t.n_primitives() * t.n_instances()

    rays = sc.c3_primary_rays(cfg, args.res, args.res)
    n = len(rays)
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    d_hits = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()

    # the headline is what a caller sees for a batch it traces for the FIRST time.  Every warm-up and timed step traces its OWN
    # batch: the C3 camera with each primary ray through a uniformly random point of its pixel (a progressive renderer's frames: the same
    # image, the same work within a fraction of a per cent, different rays every launch).  No step's rays were ever seen before, so the
    # library -- default options, as shipped -- runs each in natural claim order and records nothing; the rays come from HBM, not from a
    # buffer kept warm in the Infinity Cache.  The repeated-buffer figure of rounds 1-5 (one buffer replayed, claim order learned from its
    # earlier launches) is extras.c3_repeated_batch -> "repeated_value".
    CLOCK_WARM = 40   # untimed launches in front of the W warm-up steps (below): first launches of their own batches too, so that EVERY dispatch of the
                      # trace kernel in this command -- what `rocprofv3 --kernel-trace --stats` averages -- is what the timed steps are
    n_fresh = CLOCK_WARM + args.warmup + args.steps
    if n_fresh > 512:
        raise SystemExit("--warmup + --steps above 472: every launch traces its own 32 B/ray batch (134 MB at the default size); lower them")

    def fresh_batch(seed):
        """c3_primary_rays(jitter_seed=...) restated on the device (float64 like the numpy version; the batch used for the bit-exact check is
        read back, so nothing depends on the two agreeing to the last bit)."""
        g = torch.Generator(device="cuda")
        g.manual_seed(0xC3000 + seed)
        eye = torch.tensor(np.asarray(cfg["eye"], dtype=np.float64), device="cuda")
        f = torch.tensor(sc.normalize(np.asarray(cfg["lattice_centre"], dtype=np.float64) - np.asarray(cfg["eye"], dtype=np.float64)), device="cuda")
        r_ = torch.tensor(sc.normalize(np.cross(f.cpu().numpy(), np.array([0.0, 1.0, 0.0]))), device="cuda")
        u_ = torch.linalg.cross(r_, f)
        half = float(np.tan(np.radians(45.0) / 2))
        w = h = args.res
        px = torch.arange(w, device="cuda", dtype=torch.float64)[None, :] + torch.rand((h, w), generator=g, device="cuda", dtype=torch.float64)
        py = torch.arange(h, device="cuda", dtype=torch.float64)[:, None] + torch.rand((h, w), generator=g, device="cuda", dtype=torch.float64)
        X = (px / w * 2 - 1) * half * (w / h)
        Y = (py / h * 2 - 1) * half
        d = f[None, None, :] + X[..., None] * r_ + Y[..., None] * u_
        d = (d / d.norm(dim=-1, keepdim=True)).reshape(-1, 3)
        out = torch.empty((n, 8), dtype=torch.float32, device="cuda")
        out[:, 0:3] = eye.to(torch.float32)
        out[:, 3] = 0.0
        out[:, 4:7] = d.to(torch.float32)
        out[:, 7] = float("inf")
        return out.view(torch.uint8).reshape(-1)

    fresh = [fresh_batch(i) for i in range(n_fresh)]
    d_hits_fresh = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()

    def step(i):
        t.trace_device(fresh[i].data_ptr(), d_hits_fresh.data_ptr(), n, mode="closest", stream=stream.cuda_stream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # Untimed: bring the GPU clocks up before the W warm-up steps (a cold MI355X runs its first few dozen milliseconds of kernels
    # ~10 % slower; with small W and K that would be what gets timed).
    with rc.profile_range("headline:warmup"):
        for i in range(CLOCK_WARM):
            step(i)
        torch.cuda.synchronize()
        for i in range(args.warmup):
            step(CLOCK_WARM + i)
    # Every launch carries its own two HIP events on its kernel's dispatch (hipExtLaunchKernelGGL binds them to the kernel's start and end on
    # the launch stream): the K durations are read back after the timed region (rc_recent_kernel_ms keeps the last 47 launches').  More steps
    # than that: torch events around every step, as in rounds 1-4 -- two event packets per step, ~6 us, inside the timed region.
    # (the bound events cover the trace kernel only.  First launches enqueue nothing else; the repeated-batch extras, whose
    # launches are preceded by the two small order-rebuild kernels about one time in eight, are timed between events around the whole run.)
    own_events = args.steps <= 47
    ev = [] if own_events else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    rc.lib().rc_range_push(b"headline:timed_steps")
    if own_events:
        for i in range(args.steps):
            step(CLOCK_WARM + args.warmup + i)
    else:
        for i, (a, b) in enumerate(ev):
            a.record(stream)
            step(CLOCK_WARM + args.warmup + i)
            b.record(stream)
    # closing bracket: this rank's K steps are complete when its device has drained -- that is where its clock stops; the barrier that follows (every rank
    # has finished before anything else happens) is a collective of its own, ~1 ms even on a one-rank communicator (gpurun_out/r06camp: 5 steps measured
    # 0.73 ms per step with it inside the interval against 0.54 without), and is not part of the steps.  The job's time is the MAX over the ranks.
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    rc.lib().rc_range_pop()
    if own_events:
        per_step = t.recent_kernel_ms(args.steps)
        assert len(per_step) == args.steps and all(x > 0 for x in per_step), per_step
        launch_ms = float(np.mean(per_step))
    else:
        launch_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # what the library did with the LAST timed step (the claim-order header of the launch shape, read through the dev option the tests use):
    # "fresh" = the batch was not a repeat of a remembered one -> natural order, nothing recorded; "paused" = the shape's batches have not
    # repeated for 8 launches in a row, so its launches currently go out without even looking at their rays
    try:
        import ctypes
        hdr = torch.zeros(48, dtype=torch.int32, device="cuda")
        ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(hdr.data_ptr()), ctypes.c_void_p(t.get_option("debug_ctl_ptr")), ctypes.c_size_t(192), 3)
        hdr = hdr.cpu().numpy()
        order_state = {"last_step_fresh": int(hdr[4]), "last_step_used_a_learned_order": int(hdr[1]), "last_step_recorded": int(hdr[5]), "pause_launches_left": int(hdr[38])}
    except Exception as e:  # noqa: BLE001
        order_state = {"error": str(e)[:100]}
    last_fresh_rays = fresh[-1].cpu().numpy().view(rc.RAY_DT).copy()   # the last timed step's batch and its output: checked against the oracle below
    last_fresh_hits = d_hits_fresh.cpu().numpy().view(rc.HIT_DT).copy()
    del fresh
    t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), n, mode="closest", stream=stream.cuda_stream)  # the unjittered batch: the extras' reference output
    torch.cuda.synchronize()
    hits = d_hits.cpu().numpy().view(rc.HIT_DT)
    hit_frac = float(hits["hit"].mean())

    extras = {}

    def load_json(rel):
        try:
            return json.load(open(os.path.join(ROOT, rel)))
        except Exception:  # noqa: BLE001
            return None

    def guarded_extra(name, fn):
        """An extra measurement must never cost the headline line: failures are recorded, not raised."""
        try:
            with rc.profile_range("extra:" + name):   # roctx: `rocprofv3 --marker-trace --kernel-trace` attributes the dispatches to the extras
                fn()
        except Exception as e:  # noqa: BLE001
            extras[name + "_error"] = f"{type(e).__name__}: {e}"[:300]

    def extra_traces():
        wl_file = load_json(WORKLOADS_FILE) or {}
        wl_ok = (wl_file.get("fingerprint") or {}).get("sha256") == kernel_fingerprint()["sha256"]
        rooflines = {}
        last_ms = {}

        # launches of a batch before its steady state is measured: the reporting threshold of the learned claim order moves by a quarter per recording
        # (launches 2-4, then one in eight) and has settled after ~30 launches (profiles/r05_order_quality.txt); a render loop runs for hundreds
        STEADY, STEADY_HBM = 36, 12
        def timed(scene, rs, mode, reps=3, key=None, label=None):
            """Rate of the batch repeated on one stream.  reps <= 5: best launch by its own events (cost_order off, or one-off measurements).
            More: the batch's steady state, measured like the headline's steps -- 16 launches back to back between two events on the stream,
            two whole recording cycles of the learned claim order (one launch in 8 records costs and runs ~7 % slower, the next is preceded by
            the rebuild kernels) -- so that the figure is what a render loop averages, everything the library enqueues around the kernels
            included, not its best frame.  (Rounds 3-4 reported the mean of the launches' own kernel times, which left the three order
            dispatches of those rounds out; the kernel times still feed the per-workload rooflines.)"""
            dr = torch.from_numpy(rs.view(np.uint8).reshape(-1)).cuda()
            dh = torch.empty(len(rs) * 32, dtype=torch.uint8, device="cuda")
            ms = []
            with rc.profile_range(f"workload:{label or key or mode}:{len(rs)}x{reps}"):
                for _ in range(reps):
                    scene.trace_device(dr.data_ptr(), dh.data_ptr(), len(rs), mode=mode, stream=stream.cuda_stream)
                    ms.append(scene.last_kernel_ms())
            use = float(np.mean(ms[-8:])) if reps > 5 else min(ms)
            rate_ms = use
            if reps > 5:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with rc.profile_range(f"workload:{label or key or mode}:{len(rs)}x16 back to back"):
                    e0.record(stream)
                    for _ in range(16):
                        scene.trace_device(dr.data_ptr(), dh.data_ptr(), len(rs), mode=mode, stream=stream.cuda_stream)
                    e1.record(stream)
                    e1.synchronize()
                rate_ms = e0.elapsed_time(e1) / 16
            if key:
                last_ms[key] = {"mean_of_last_8_ms": round(float(np.mean(ms[-8:])), 4), "best_ms": round(min(ms), 4), "launches": reps, "back_to_back_ms": round(rate_ms, 4)}
                r = make_workload_roofline((wl_file.get("workloads") or {}).get(key), use, len(rs), wl_ok)
                rooflines[key] = r if r else {"frac": None, "note": f"{WORKLOADS_FILE} missing, without this workload, or captured from other kernel sources (fingerprint)"}
            return round(len(rs) / rate_ms / 1e3, 1)
        def in_flight(scene, rs, n_streams=4, batches=32):
            """Rate with n_streams launches of the same batch in flight (one stream and one output each): what a caller that pipelines
            mid-size batches sees -- the tail of one launch (waves waiting for their longest rays) overlaps the bulk of the next."""
            dr = torch.from_numpy(rs.view(np.uint8).reshape(-1)).cuda()
            sts = [torch.cuda.Stream() for _ in range(n_streams)]
            outs = [torch.empty(len(rs) * 32, dtype=torch.uint8, device="cuda") for _ in range(n_streams)]
            rate = 0.0
            for _ in range(2):
                torch.cuda.synchronize()
                p0 = time.perf_counter()
                for i in range(batches):
                    scene.trace_device(dr.data_ptr(), outs[i % n_streams].data_ptr(), len(rs), stream=sts[i % n_streams].cuda_stream)
                torch.cuda.synchronize()
                rate = round(len(rs) * batches / (time.perf_counter() - p0) / 1e6, 1)
            return rate
        # Throughput with consecutive batches overlapped on two streams (fill / drain of one launch hidden behind the next; each
        # stream has its own stack spill region, each launch its own counters).  Not the headline: per-launch attribution is lost.
        def pipelined(kernel_opt):
            t.set_option("kernel", kernel_opt)
            st2 = [torch.cuda.Stream(), torch.cuda.Stream()]
            dh2 = [torch.empty(n * 32, dtype=torch.uint8, device="cuda") for _ in range(2)]
            for j in (0, 1):
                t.trace_device(d_rays.data_ptr(), dh2[j].data_ptr(), n, stream=st2[j].cuda_stream)
            torch.cuda.synchronize()
            k = 40
            p0 = time.perf_counter()
            for i in range(k):
                t.trace_device(d_rays.data_ptr(), dh2[i % 2].data_ptr(), n, stream=st2[i % 2].cuda_stream)
            torch.cuda.synchronize()
            rate = round(n * k / (time.perf_counter() - p0) / 1e6, 1)
            t.set_option("kernel", -1)
            return rate
        extras["c3_two_stream_pipelined_mrays_s"] = {"kernel3": pipelined(3), "kernel5": pipelined(5)}
        # the headline batch with the entry cull switched off (same kernel otherwise): what the cull is worth, and a bytewise comparison
        t.set_option("entry_cull", 0)
        dh_off = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
        best_off = 1e30
        for _ in range(4):
            t.trace_device(d_rays.data_ptr(), dh_off.data_ptr(), n, stream=stream.cuda_stream)
            best_off = min(best_off, t.last_kernel_ms())
        torch.cuda.synchronize()
        t.set_option("entry_cull", 1)
        extras["c3_entry_cull_off"] = {"mrays_s": round(n / best_off / 1e3, 1), "hits_identical_to_the_default_run": bool(torch.equal(dh_off, d_hits))}
        # the headline repeats ONE ray buffer, so its claim order is learned from bit-identical rays.  (a) the same batch with the
        # order switched off (what a never-seen batch gets); (b) a camera that moves a little every frame: 16 different batches, eye shifted
        # by 0.02 per frame -- the device recognises them as the same batch (sample rays compared) and keeps reusing / refreshing ONE order
        def back_to_back(buffers, out_buf, rounds=1):
            """ms per launch of the given ray buffers traced one after the other on the stream, between two events: measured like the headline's steps"""
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(rounds):
                for b in buffers:
                    t.trace_device(b.data_ptr(), out_buf.data_ptr(), n, stream=stream.cuda_stream)
            e1.record(stream)
            e1.synchronize()
            return e0.elapsed_time(e1) / (rounds * len(buffers))
        # the figure rounds 1-5 headlined: ONE buffer replayed, claim order learned from its earlier launches (a render loop with a still
        # camera, repeated queries).  64 launches for the pause the headline's fresh batches caused + 120 to learn (the reporting threshold needs ~80 launches to settle on this batch), then K back to back between two events
        # -- the order-rebuild kernel pairs that precede about one launch in eight are inside the interval.
        for _ in range(64 + 120):   # the headline's never-repeating batches have put the launch shape into its 64-launch pause (order_commit): sit that out first; then ~80 launches until the reporting threshold has settled (tools/probes/repeat_vs_natural_probe.py)
            t.trace_device(d_rays.data_ptr(), dh_off.data_ptr(), n, stream=stream.cuda_stream)
        b2b_rep = back_to_back([d_rays], dh_off, rounds=args.steps)
        try:
            import ctypes
            hdr = torch.zeros(48, dtype=torch.int32, device="cuda")
            ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(hdr.data_ptr()), ctypes.c_void_p(t.get_option("debug_ctl_ptr")), ctypes.c_size_t(192), 3)
            hdr = hdr.cpu().numpy()
            rep_state = {"last_launch_fresh": int(hdr[4]), "last_launch_used_a_learned_order": int(hdr[1]), "last_launch_recorded": int(hdr[5]), "pause_launches_left": int(hdr[38]), "slot_generation": int(hdr[12 + int(hdr[0])])}
        except Exception as e:  # noqa: BLE001
            rep_state = {"error": str(e)[:100]}
        extras["c3_repeated_batch"] = {"mrays_s": round(n / b2b_rep / 1e3, 1), "launches": args.steps, "hits_identical_to_the_first_launch": bool(torch.equal(dh_off, d_hits)), "order_state": rep_state,
                                       "note": "the unjittered C3 batch replayed from one buffer with the claim order learned from its own earlier launches (option cost_order, default on): "
                                               "what rounds 1-5 reported as `value`; needs an identical previous launch"}
        t.set_option("cost_order", 0)
        ms_off = []
        for _ in range(8):
            t.trace_device(d_rays.data_ptr(), dh_off.data_ptr(), n, stream=stream.cuda_stream)
            ms_off.append(t.last_kernel_ms())
        b2b_off = back_to_back([d_rays], dh_off, rounds=args.steps)
        t.set_option("cost_order", 1)
        extras["c3_cost_order_off"] = {"mrays_s": round(n / b2b_off / 1e3, 1), "kernel_mean_mrays_s": round(n / float(np.mean(ms_off)) / 1e3, 1), "kernel_best_mrays_s": round(n / min(ms_off) / 1e3, 1),
                                       "hits_identical_to_the_default_run": bool(torch.equal(dh_off, d_hits)),
                                       "note": f"natural claim order -- what a batch traced for the first time gets: {args.steps} launches back to back between two events, like the headline's steps "
                                               "(kernel_mean: the mean of 8 single launches' own events with the host waiting in between, rounds 3-4's figure; the rays of this one buffer are cache-warm "
                                               "either way -- c3_moving_camera has rays that are new every launch)"}
        frames = [torch.from_numpy(sc.pinhole_rays(args.res, args.res, cfg["eye"] + np.array([0.02 * k, 0.01 * k, 0.0]), cfg["lattice_centre"], 45.0).view(np.uint8).reshape(-1)).cuda() for k in range(16)]
        for f in frames:   # (frames 1-16: the shape's first launches of these batches, eight of them inside the mechanism)
            t.trace_device(f.data_ptr(), dh_off.data_ptr(), n, stream=stream.cuda_stream)
        on_ms, off_ms = [], []
        for _ in range(3):   # interleaved: this box's launch times drift by a per cent or two within a process
            on_ms.append(back_to_back(frames, dh_off, rounds=1))
            t.set_option("cost_order", 0)
            off_ms.append(back_to_back(frames, dh_off, rounds=1))
            t.set_option("cost_order", 1)
        b2b_j, b2b_j0 = float(np.mean(on_ms)), float(np.mean(off_ms))
        extras["c3_moving_camera"] = {"mrays_s": round(n / b2b_j / 1e3, 1), "frames": 48, "mrays_s_cost_order_off": round(n / b2b_j0 / 1e3, 1),
                                      "note": "every launch traces DIFFERENT rays (the eye moves 0.022 per frame, 16 ray buffers of 134 MB in rotation: unlike the headline's one buffer they "
                                              "do not stay in the Infinity Cache); frames 17-112 in six runs of 16 back to back between two events, alternately with cost_order 1 and 0 (48 frames each).  A moving camera's frames are recognised as "
                                              "the batch of the frame before but are not REPEATS of it: after eight such launches the shape's next 64 launches go out outside the mechanism "
                                              "(round 5; an order learned from similar rays gains less than its recording launches cost: docs/EXPERIMENTS.md), so the two figures agree "
                                              "within noise; the first eight frames -- not in the figure -- pay ~2 %"}
        del dh_off, frames
        shadow = sc.c3_shadow_rays(cfg, rays, hits)
        extras["c3_any_hit_shadow_mrays_s"] = timed(t, shadow, "any", reps=STEADY, key="shadow")
        t.set_option("cost_order", 0)
        extras["c3_any_hit_shadow_first_launch_mrays_s"] = timed(t, shadow, "any", reps=5, label="shadow_first_launch")
        t.set_option("cost_order", 1)
        bounce = sc.c4_bounce_rays(cfg, rays, hits, 4 * n)
        extras["c4_incoherent_16M_closest_mrays_s"] = timed(t, bounce, "closest", reps=STEADY, key="c4")
        del bounce, shadow
        # A path tracer's bounce rays are NEW every launch: six different 4.19 M-ray incoherent batches in rotation, three rounds.  No batch
        # comes back before its slot has been given away, so nothing is ever learned -- and nothing must be paid for trying: a batch seen for
        # the first time does not record (recording costs ~7 % of a launch).
        fresh = [torch.from_numpy(sc.c4_bounce_rays(cfg, rays, hits, n, seed=0xC400 + k).view(np.uint8).reshape(-1)).cuda() for k in range(6)]
        dh_f = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
        nr = {}
        t.set_option("cost_order", 0)
        for f in fresh * 2:  # untimed: first touches of six new 134 MB buffers run 5-10 % slower than later passes, whatever the option
            t.trace_device(f.data_ptr(), dh_f.data_ptr(), n, stream=stream.cuda_stream)
        t.set_option("cost_order", 1)
        for f in fresh * 2:  # untimed too: the shape's first launches under the mechanism (the eighth starts the pause)
            t.trace_device(f.data_ptr(), dh_f.data_ptr(), n, stream=stream.cuda_stream)
        ms = {1: [], 0: []}
        for rep in range(4):  # interleaved (launch times drift by a per cent or two within a process): 6 launches back to back per option and round
            for co in (1, 0):
                t.set_option("cost_order", co)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for f in fresh:
                    t.trace_device(f.data_ptr(), dh_f.data_ptr(), n, stream=stream.cuda_stream)
                e1.record(stream)
                e1.synchronize()
                ms[co].append(e0.elapsed_time(e1) / len(fresh))
        for co in (1, 0):
            nr[f"cost_order_{co}_mrays_s"] = round(n / float(np.mean(ms[co])) / 1e3, 1)
        t.set_option("cost_order", 1)
        nr["note"] = ("4.19 M incoherent bounce rays, a different batch every launch (6 in rotation > the 4 batch slots): 24 launches per option, back to back in runs of 6, the options alternating; "
                      "a batch's first launch records nothing, and after eight launches that repeated no remembered batch the shape's next 64 launches go out in natural order without looking at their "
                      "rays (counted and decided inside the launches): the two figures should agree within noise")
        extras["c4_never_repeating_batches"] = nr
        del fresh, dh_f
        mid = sc.c3_primary_rays(cfg, 1024, 1024)
        extras["c3_1Mi_primary_closest_mrays_s"] = timed(t, mid, "closest", reps=STEADY, key="c3")
        cfg2 = sc.config_c2()
        t2 = rc.TLAS(local_rank)
        t2.add_geometry(*cfg2["blas"][0])
        t2.push_instances(1, cfg2["instances"][0][1], cfg2["instances"][0][2])
        t2.sync()
        rays2 = rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"])
        # mid-size batches: `timed` repeats the batch and keeps the best launch, i.e. the steady state of a render loop -- from the second launch
        # on the chunks are claimed in the order learned from the launch before (cost-ordered claiming).  A batch traced for the
        # first time has nothing to go by: that rate is measured with the option off.
        t2.set_option("cost_order", 0)
        extras["c2_100k_blas_1M_coherent_closest_first_launch_mrays_s"] = timed(t2, rays2, "closest", reps=5, label="c2_first_launch")
        t2.set_option("cost_order", 1)
        extras["c2_100k_blas_1M_coherent_closest_mrays_s"] = timed(t2, rays2, "closest", reps=STEADY, key="c2")
        extras["c2_100k_blas_1M_coherent_closest_4_in_flight_mrays_s"] = in_flight(t2, rays2)
        # four INDEPENDENT C2-sized batches (four view directions) handed over in ONE call (rc_trace_closest_device_batches: round-robin
        # over the scene's auxiliary streams, forked from and joined into the caller's stream).  first_call: batches never seen; steady: the same four
        # batches again (each auxiliary stream's history has learned its batch), 8 calls back to back between two events on the caller's stream.
        dirs4 = [cfg2["viewdir"], (-0.8, 0.4, 0.3), (0.1, -1.0, 0.25), (0.6, 0.6, -0.5)]
        b4 = [rc.generate_ray_grid(t2, d, cfg2["grid"]) for d in dirs4]
        d4 = [torch.from_numpy(r.view(np.uint8).reshape(-1)).cuda() for r in b4]
        h4 = [torch.empty(len(r) * 32, dtype=torch.uint8, device="cuda") for r in b4]
        h4_single = [torch.empty(len(r) * 32, dtype=torch.uint8, device="cuda") for r in b4]
        for i in range(4):
            t2.trace_device(d4[i].data_ptr(), h4_single[i].data_ptr(), len(b4[i]), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        def one_call():
            t2.trace_device_batches([x.data_ptr() for x in d4], [x.data_ptr() for x in h4], [len(r) for r in b4], stream=stream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); one_call(); e1.record(stream); e1.synchronize()
        first_call_ms = e0.elapsed_time(e1)
        same4 = all(bool(torch.equal(h4[i], h4_single[i])) for i in range(4))
        for _ in range(24):
            one_call()
        e0.record(stream)
        for _ in range(8):
            one_call()
        e1.record(stream); e1.synchronize()
        steady_ms = e0.elapsed_time(e1) / 8
        # the same four batches one after the other on the stream (what the single-batch API gives), steady state
        for _ in range(24):
            for i in range(4):
                t2.trace_device(d4[i].data_ptr(), h4_single[i].data_ptr(), len(b4[i]), stream=stream.cuda_stream)
        e0.record(stream)
        for _ in range(8):
            for i in range(4):
                t2.trace_device(d4[i].data_ptr(), h4_single[i].data_ptr(), len(b4[i]), stream=stream.cuda_stream)
        e1.record(stream); e1.synchronize()
        serial_ms = e0.elapsed_time(e1) / 8
        nr4 = sum(len(r) for r in b4)
        extras["c2_4_independent_batches_one_call"] = {"rays": nr4, "first_call_mrays_s": round(nr4 / first_call_ms / 1e3, 1), "mrays_s": round(nr4 / steady_ms / 1e3, 1),
                                                       "one_after_the_other_mrays_s": round(nr4 / serial_ms / 1e3, 1), "hits_identical_to_single_calls": same4,
                                                       "note": "four 1 M-ray batches (four view directions of the C2 scene) in one rc_trace_closest_device_batches call vs four "
                                                               "rc_trace_closest_device calls in a row on one stream; timed between two events on the caller's stream"}
        del b4, d4, h4, h4_single
        # VERDICT r3 #5a: consecutive same-size batches that DIFFER -- two view directions alternating on one stream.  The device tells the
        # batches apart by their sample rays and keeps an order for each; with the option off both run in natural order.
        rays2b = rc.generate_ray_grid(t2, (-0.8, 0.4, 0.3), cfg2["grid"])
        da, db = (torch.from_numpy(r.view(np.uint8).reshape(-1)).cuda() for r in (rays2, rays2b))
        dh2 = torch.empty(len(rays2) * 32, dtype=torch.uint8, device="cuda")
        alt = {}
        for co in (0, 1):
            t2.set_option("cost_order", co)
            ms = []
            for k in range(24):
                t2.trace_device((da if k % 2 == 0 else db).data_ptr(), dh2.data_ptr(), len(rays2), stream=stream.cuda_stream)
                ms.append(t2.last_kernel_ms())
            alt[f"cost_order_{co}_mean_ms"] = round(float(np.mean(ms[8:])), 4)
        t2.set_option("cost_order", 1)
        alt["learned_vs_natural"] = round(alt["cost_order_1_mean_ms"] / alt["cost_order_0_mean_ms"], 4)
        alt["note"] = "C2, two different 1 M-ray batches A B A B ... on one stream, mean of launches 9..24; < 1: the learned order wins although consecutive launches differ"
        extras["c2_alternating_batches"] = alt
        del da, db, dh2
        t2.free()
        # Top levels beyond the 256 instances the full LDS kernel takes (kernel 6: TLAS / BLAS tops in LDS, the rest from memory):
        # the C3 BLAS on bigger lattices, same 4 M-ray pinhole camera
        big = {}
        for lattice in ((10, 10, 5), (20, 20, 12)):
            cfgb = sc.config_c3(lattice=lattice)
            tb = rc.TLAS(local_rank)
            for verts, meta in cfgb["blas"]:
                tb.add_geometry(verts, meta)
            for b, xf, ids in cfgb["instances"]:
                tb.push_instances(b, xf, ids)
            tb.sync()
            big[str(int(np.prod(lattice)))] = {"triangles": tb.n_primitives() * tb.n_instances(),
                                               "mrays_s": timed(tb, sc.c3_primary_rays(cfgb, args.res, args.res), "closest", reps=STEADY, label=f"c3_{int(np.prod(lattice))}_instances"),
                                               "tlas_top_k": tb.get_option("tlas_top_k"), "blas_top_k": tb.get_option("blas_top_k")}
            tb.free()
        extras["c3_blas_more_instances_closest"] = big
        # The reference's own published traversal benchmark shape (benchmarks/implicitbvh_comparison.md:37-39): random geometry in one
        # BLAS, 1 M rays closest_hit -- 8.99 / 11.08 / 15.41 ms for 250 k / 1 M / 4 M triangles on an RX 7900 XTX (ray distribution
        # unstated there; here the coherent 1000 x 1000 grid of get_illumination).
        ref = {}
        for nt, ref_ms in ((250_000, 8.99), (1_000_000, 11.08), (4_000_000, 15.41)):
            tb = rc.TLAS(local_rank)
            dv = torch.from_numpy(sc.random_triangles(nt, 42, edge=0.01)).cuda()
            tb.add_geometry_device(dv.data_ptr(), nt)
            tb.push_instances(1)
            tb.sync()
            del dv
            rg = rc.generate_ray_grid(tb, (0.3, 0.2, 1.0), 1000)
            tb.set_option("cost_order", 0)
            first = timed(tb, rg, "closest", reps=3, label=f"random_{nt}_first_launch")
            tb.set_option("cost_order", 1)
            rate = timed(tb, rg, "closest", reps=STEADY, key="r1m" if nt == 1_000_000 else None, label=f"random_{nt}")
            ref[str(nt)] = {"mrays_s": rate, "first_launch_mrays_s": first, "ms_per_1M_rays": round(1e3 / rate, 3), "mrays_s_4_in_flight": in_flight(tb, rg), "reference_rx7900xtx_ms": ref_ms}
            tb.free()
        extras["random_geometry_1M_rays_closest"] = ref
        torch.cuda.empty_cache()
        # An HBM-bound regime (the only place BASELINE's "HBM roofline" wording is testable): a 4 M-triangle BLAS -- a 512 MB node array,
        # far beyond L2 + Infinity Cache -- and 4 M incoherent rays.  Rate and node fetches (the product's instrumented kernel) are
        # measured here; the physical HBM bytes per launch come from the workload's FETCH_SIZE / WRITE_SIZE passes in WORKLOADS_FILE (make_hbm_regime).
        # A second point past the Infinity Cache: a 16 M-triangle BLAS -- 2 GB of nodes, 8 x the 256 MiB MALL -- under the same
        # rays: there FETCH_SIZE is DRAM traffic to within the MALL's small share of hits.
        g = np.random.default_rng(7)
        ro = g.random((n, 3))
        rd = g.standard_normal((n, 3))
        rd /= np.linalg.norm(rd, axis=1, keepdims=True)
        inc = sc.make_rays(ro, rd)
        del ro, rd
        for key, nt, name in (("hbm", 4_000_000, "hbm_regime_4M_tris_4M_incoherent_rays"), ("hbm16", 16_000_000, "hbm_regime_16M_tris_4M_incoherent_rays")):
            tb = rc.TLAS(local_rank)
            dv = torch.from_numpy(sc.random_triangles(nt, 42, edge=0.01)).cuda()
            tb.add_geometry_device(dv.data_ptr(), nt)
            tb.push_instances(1)
            tb.sync()
            del dv
            rate = timed(tb, inc, "closest", reps=STEADY_HBM, label=f"hbm_regime_{nt // 1_000_000}M_tris", key=key)
            hbm_launch_ms = last_ms[key]["mean_of_last_8_ms"]
            tb.set_option("kernel", 3); tb.set_option("stats", 1)
            timed(tb, inc, "closest", reps=1, label=f"hbm_regime_{nt // 1_000_000}M_tris_stats_kernel3")
            st = [tb.get_option(f"stat{i}") for i in range(8)]
            tb.set_option("stats", 0); tb.set_option("kernel", -1)
            tree_bytes = int(tb.adapt().n_blas_nodes) * 64 if hasattr(tb.adapt(), "n_blas_nodes") else (2 * nt - 1) * 64
            tb.free()
            fetches = (st[3] + st[5]) / n + 1.0
            extras[name] = make_hbm_regime(rate, fetches, n, hbm_launch_ms, (wl_file.get("workloads") or {}).get(key), wl_ok, key=key, tree_bytes=tree_bytes)
            torch.cuda.empty_cache()
        del inc
        torch.cuda.empty_cache()
        extras["rooflines"] = {"note": "per extra workload: VALU issue against the guide's 2-cycle peak from the per-launch counters of " + WORKLOADS_FILE +
                                       " and the launch time measured in this run (the workload's steady-state mean); recomputed by tests/test_bench_roofline.py",
                               "workloads": rooflines, "launch_ms": last_ms}


    def extra_builds():
        # LBVH build (row a14): device pipeline time for triangle soup already in HBM; the reference publishes
        # 4.93 / 7.46 / 16.16 ms for 250k / 1M / 4M triangles on an RX 7900 XTX (benchmarks/implicitbvh_comparison.md:12-14).
        builds = {}
        for nt, ref_ms in ((250_000, 4.93), (1_000_000, 7.46), (4_000_000, 16.16)):
            dv = torch.from_numpy(sc.random_triangles(nt, 42, edge=0.01)).cuda()
            tb = rc.TLAS(local_rank)
            best = 1e30
            for _ in range(3):
                tb.add_geometry_device(dv.data_ptr(), nt)
                best = min(best, tb.last_kernel_ms())
            builds[str(nt)] = {"ms": round(best, 3), "Mtris_s": round(nt / best / 1e3, 1), "reference_rx7900xtx_ms": ref_ms}
            tb.free()
            del dv
        extras["blas_build_device"] = builds
        torch.cuda.empty_cache()
        # Dynamic scenes (rows a15 / f2; test/test_tlas_stress.jl:284-327 moves 5 000 instances per frame): update_transforms! + sync!
        # = upload of the transforms, inverses, instance boxes, TLAS refit in place, traversal records -- host wall time per frame --
        # and the device-resident form (descriptors rewritten in HBM by the caller, rc_refit_device: no PCIe traffic).
        g = np.random.default_rng(5)
        n_inst = 5000
        xf = np.tile(sc.IDENTITY3x4, (n_inst, 1)).astype(np.float32)
        xf[:, [3, 7, 11]] = (g.random((n_inst, 3)) * 40).astype(np.float32)
        td = rc.TLAS(local_rank)
        hnd = td.push(sc.fan_sphere(16, 9), xf)
        td.sync()
        frames = []
        for f in range(12):
            xf[:, 3] += 0.01
            f0 = time.perf_counter()
            td.update_transforms(hnd, xf)
            td.sync()
            td.wait_for_gpu()
            frames.append(time.perf_counter() - f0)
            assert td.last_sync_action == "refit"
        dev = []
        for f in range(12):
            f0 = time.perf_counter()
            td.refit_device(recompute_inverse=True)
            td.wait_for_gpu()
            dev.append(time.perf_counter() - f0)
        extras["tlas_refit_5000_instances"] = {"update_transforms_plus_sync_ms": round(min(frames) * 1e3, 3), "refit_device_ms": round(min(dev) * 1e3, 3),
                                               "refit_device_kernels_ms": round(td.last_kernel_ms(), 3)}
        td.free()


    def extra_bvh4_and_collision():
        # BVH4 (row a16): collapse time and closest_hit4 rate on C2's 100k-triangle BLAS, same coherent 1 M rays as the BVH2 extra
        cfg2 = sc.config_c2()
        t2 = rc.TLAS(local_rank)
        t2.add_geometry(*cfg2["blas"][0])
        t2.push_instances(1, cfg2["instances"][0][1], cfg2["instances"][0][2])
        t2.sync()
        rs = rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"])
        t2.free()
        b4 = rc.build_blas4(*cfg2["blas"][0], device=local_rank)
        collapse_ms = b4.last_kernel_ms()
        import ctypes
        for _ in range(2):  # the first collapse pays for the scratch allocations
            rc._capi.check(rc.lib().rc_blas4_build(b4._scene._h, b4._blas_id, ctypes.byref(ctypes.c_uint32())))
            collapse_ms = min(collapse_ms, b4.last_kernel_ms())
        dr = torch.from_numpy(rs.view(np.uint8).reshape(-1)).cuda()
        dh = torch.empty(len(rs) * 32, dtype=torch.uint8, device="cuda")
        best = 1e30
        for _ in range(5):
            b4.trace_device(dr.data_ptr(), dh.data_ptr(), len(rs), stream=stream.cuda_stream)
            best = min(best, b4.last_kernel_ms())
        extras["bvh4_c2"] = {"collapse_ms": round(collapse_ms, 3), "nodes4": int(b4.num_interior),
                             "closest_hit4_1M_coherent_mrays_s": round(len(rs) / best / 1e3, 1)}
        # collision broad phase (src/collision.jl) on 5000 instances
        g = np.random.default_rng(11)
        xf = np.tile(sc.IDENTITY3x4, (5000, 1))
        xf[:, [3, 7, 11]] = (g.random((5000, 3)) * 20).astype(np.float32)
        tc = rc.TLAS(local_rank)
        tc.push(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 1, 1, 0, 1, 0, 1, 1]], np.float32), xf)
        res = rc.collide_instances(tc)
        extras["collide_instances_5000"] = {"pairs": res.num_contacts, "device_ms": round(tc.last_kernel_ms(), 3)}
        tc.free()


    def extra_view_factors():
        # view_factors (BASELINE config C5: ~50k-triangle closed scene, rays_per_triangle = 4096 => 204.9 M rays, N x N
        # u32 = 10 GB) through the multi-GPU driver: every rank takes part; both partitions are timed.
        from raycore_jl_amd import distributed as rd
        cfg5 = sc.config_c5()
        v0 = time.perf_counter()
        t5 = rc.TLAS(local_rank)
        t5.add_geometry(*cfg5["blas"][0])
        t5.push_instances(1, cfg5["instances"][0][1], cfg5["instances"][0][2])
        t5.sync()
        replicate_ms = (time.perf_counter() - v0) * 1e3   # every rank uploads and builds its own copy of the scene, side by side
        n5, rpt = t5.n_primitives(), cfg5["rays_per_triangle"]
        vf = {"n_prims": n5, "rays_per_triangle": rpt, "n_rays": n5 * rpt, "matrix_bytes": 4 * n5 * n5}
        t5._prims()  # metadata read-back outside the timed region
        for mode in ("rows_sharded", "rows", "rays"):
            rd.view_factors_distributed(t5, 4, 7, mode=mode)  # warm-up: allocator, RCCL channels
            fence()
            v0 = time.perf_counter()
            m = rd.view_factors_distributed(t5, rpt, 7, mode=mode)
            fence()
            vdt = time.perf_counter() - v0
            if mode == "rows_sharded":  # result stays row-sharded: count = sum over ranks
                cnt = m[0].sum(dtype=torch.int64)
                if use_dist and args.backend == "nccl":
                    dist.all_reduce(cnt)
                counted = int(cnt.item())
            else:
                counted = int(m.sum(dtype=torch.int64).item()) if rank == 0 else 0
            if rank == 0:
                # seconds = end to end on the slowest rank, INCLUDING the allocation and zero fill of this rank's block of the matrix
                # (10 GB at N = 1, fresh from the driver each time because the cache is emptied between modes) and any gather / reduce;
                # kernel_ms = rank 0's trace kernel alone
                vf[mode] = {"seconds": round(vdt, 4), "Mrays_s": round(n5 * rpt / vdt / 1e6, 1), "counted": counted,
                            "kernel_ms_rank0": round(t5.last_kernel_ms(), 3)}
            del m
            torch.cuda.empty_cache()
        # The per-triangle totals (column / row sums of the matrix; what the reference's users read off it) with the RAYS sharded over the
        # ranks and ONE reduce of 2 N int64 (RCCL ncclReduce over xGMI when world > 1): no N x N array, no PCIe floor.
        fence()
        v0 = time.perf_counter()
        rd.view_factor_totals_distributed(t5, 4, 7)       # the first reduce of the process group: RCCL's channel / xGMI connection set-up
        fence()
        first_reduce_ms = (time.perf_counter() - v0) * 1e3
        best_tot, tot = 1e30, None
        for _ in range(3):
            fence()
            v0 = time.perf_counter()
            tot = rd.view_factor_totals_distributed(t5, rpt, 7)
            fence()
            best_tot = min(best_tot, time.perf_counter() - v0)
        if rank == 0:
            # the one-GPU answer, from this rank alone (40 ms): every sharded mode above must have counted exactly these rays
            one_r, one_e = rc.view_factor_totals(t5, rpt, 7)
            one_count = int(one_r.sum())
            vf["one_gpu_count"] = one_count
            vf["rccl_ranks"] = world if (use_dist and args.backend == "nccl") else 0
            for mode in ("rows_sharded", "rows", "rays"):
                vf[mode]["equals_one_gpu"] = vf[mode]["counted"] == one_count
                assert vf[mode]["equals_one_gpu"], f"view_factors mode {mode} on {world} ranks counted {vf[mode]['counted']} rays, one GPU counts {one_count}"
            assert np.array_equal(tot[0], one_r) and np.array_equal(tot[1], one_e), "rays-sharded totals differ from the one-GPU totals"
            vf["totals_rays_sharded"] = {"seconds": round(best_tot, 4), "Mrays_s": round(n5 * rpt / best_tot / 1e6, 1), "counted": int(tot[0].sum()), "ranks": world,
                                         "reduce_bytes": 16 * n5, "kernel_ms_rank0": round(t5.last_kernel_ms(), 3),
                                         # the fixed costs, itemised and OUTSIDE `seconds`
                                         "totals_ms": round(best_tot * 1e3, 3), "comm_init_ms": round(comm_init_ms, 2), "replicate_ms": round(replicate_ms, 2),
                                         "first_reduce_ms": round(first_reduce_ms, 2), "rccl_ranks": world if (use_dist and args.backend == "nccl") else 0,
                                         "note": "received[N] + emitted[N] (u64) accumulated on the device, rays sharded over the ranks, one reduce; equals the matrix's column / row sums"}
            vf["totals_rays_sharded"]["equals_one_gpu"] = True  # (asserted above, element by element)
        if world == 1:
            rc.view_factor_totals(t5, 4, 7)
            rc.view_factor_totals(t5, rpt, 7)
            vf["totals_ms"] = round(t5.last_kernel_ms(), 3)
        # What the API returns: a HOST N x N matrix (src/kernels.jl:74-78).  One process per GPU: every rank copies its block of rows
        # into a matrix in shared memory over its own PCIe link (distributed.view_factors_host_matrix); the time includes creating and
        # faulting in the 10 GB matrix.  One GPU: rc_view_factors, row chunks traced while the finished ones travel.
        rd.view_factors_host_matrix(t5, 4, 7)
        fence()
        v0 = time.perf_counter()
        hm = rd.view_factors_host_matrix(t5, rpt, 7)
        fence()
        vdt = time.perf_counter() - v0
        counted_fresh = int(np.asarray(hm).sum(dtype=np.int64)) if rank == 0 else 0
        del hm
        # a solver calls view_factors again and again: the shared matrix is created once (SharedHostMatrix) and refilled
        shared = rd.SharedHostMatrix(n5)
        best = 1e30
        for _ in range(2):
            fence()
            v0 = time.perf_counter()
            hm = rd.view_factors_host_matrix(t5, rpt, 7, out=shared)
            fence()
            best = min(best, time.perf_counter() - v0)
        if rank == 0:
            vf["host_matrix_shared_memory"] = {"fresh_matrix_s": round(vdt, 4), "reused_matrix_s": round(best, 4), "Mrays_s_reused": round(n5 * rpt / best / 1e6, 1), "ranks": world,
                                               "GBs_into_host_memory_reused": round(4 * n5 * n5 / best / 1e9, 1), "counted": counted_fresh,
                                               "counted_reused": int(np.asarray(hm).sum(dtype=np.int64)),
                                               "note": "one process per GPU, every rank writes its rows into a matrix in /dev/shm over its own PCIe link; fresh = incl. creating + "
                                                       "faulting in the matrix (shared-memory pages: ~0.3 s per GB from one rank), reused = a SharedHostMatrix created once"}
        del shared
        del hm
        if world == 1:
            out_m = np.empty((n5, n5), dtype=np.uint32, order="F")
            e2e = {}
            v0 = time.perf_counter()
            rc.view_factors(t5, rpt, 7, out=out_m)  # first call: the matrix's pages are faulted in inside the call (parallel MADV_POPULATE_WRITE)
            e2e["fresh_matrix_s"] = round(time.perf_counter() - v0, 4)
            for key, register in (("reused_matrix_s", False), ("reused_registered_matrix_s", True)):
                if register:
                    t5.host_register(out_m)
                best = 1e30
                for _ in range(2):
                    v0 = time.perf_counter()
                    rc.view_factors(t5, rpt, 7, out=out_m)
                    best = min(best, time.perf_counter() - v0)
                e2e[key] = round(best, 4)
                if register:
                    t5.host_unregister(out_m)
            e2e["device_pipeline_ms"] = round(t5.last_kernel_ms(), 3)
            e2e["counted"] = int(out_m.sum(dtype=np.int64))
            e2e["pcie_floor_s_at_56_GBs"] = round(4 * n5 * n5 / 56e9, 4)
            vf["host_matrix_e2e_s"] = e2e["reused_matrix_s"]
            vf["host_matrix_e2e"] = e2e
            del out_m
        t5.free()
        if rank == 0:
            extras["view_factors_c5"] = vf

    def extra_one_process_multi_device():
        """The C ABI's own multi-device entry points (one process, one scene per device) -- only measurable where this
        process sees more than one GPU (the driver's 8-GPU node at N = 1).  RC_BENCH_FORCE_MULTI=k puts k replicas on device 0 instead:
        a functional run of the same code, not a scaling number."""
        forced = int(os.environ.get("RC_BENCH_FORCE_MULTI", "0"))
        n_dev = rc.device_count()
        devices = [0] * forced if forced > 1 else list(range(n_dev))
        if len(devices) < 2:
            return
        md = {"devices": devices, "forced_replicas_on_one_device": bool(forced > 1)}
        # (1) view_factors into a host matrix, ROWS: every device brings its row block home over its own PCIe link, no collective
        cfg5 = sc.config_c5()
        scenes5 = []
        v0 = time.perf_counter()
        for d in devices:
            t = rc.TLAS(d)
            t.add_geometry(*cfg5["blas"][0])
            t.push_instances(1, cfg5["instances"][0][1], cfg5["instances"][0][2])
            scenes5.append(t.sync())
        replicate_ms = (time.perf_counter() - v0) * 1e3                # upload + BLAS / TLAS build of the scene, once per device, one after the other
        # the fixed costs of a set of devices, outside every timed call below and itemised: RCCL + communicator, streams, staging, warm-up collective
        prep = rc.multi_prepare(scenes5)
        md["fixed_costs_ms"] = {"replicate_scene_on_all_devices": round(replicate_ms, 2), "comm_init": round(prep["comm_init_ms"], 2),
                                "streams_and_buffers": round(prep["streams_and_buffers_ms"], 2), "warmup_collective": round(prep["warmup_collective_ms"], 2),
                                "rccl_ranks": prep["rccl_ranks"]}
        n5, rpt = scenes5[0].n_primitives(), cfg5["rays_per_triangle"]
        out_m = np.empty((n5, n5), dtype=np.uint32, order="F")
        rc.view_factors(scenes5[0], rpt, 7, out=out_m)            # one device; also faults the matrix in
        want = int(out_m.sum(dtype=np.int64))
        best1 = bestn = 1e30
        for _ in range(2):
            v0 = time.perf_counter(); rc.view_factors(scenes5[0], rpt, 7, out=out_m); best1 = min(best1, time.perf_counter() - v0)
        for _ in range(2):
            v0 = time.perf_counter(); rc.view_factors_multi(scenes5, rpt, 7, mode="rows", out=out_m); bestn = min(bestn, time.perf_counter() - v0)
        import zlib
        md["view_factors_c5_host_matrix"] = {"one_device_s": round(best1, 4), "all_devices_rows_s": round(bestn, 4), "speedup": round(best1 / bestn, 2),
                                             "same_count": int(out_m.sum(dtype=np.int64)) == want, "matrix_bytes": 4 * n5 * n5}
        assert md["view_factors_c5_host_matrix"]["same_count"], "ROWS over all devices counted different rays than one device"
        # (1b) the per-triangle totals: rays sharded over the devices, ONE ncclReduce of 2 N u64 over xGMI (host sum for forced replicas)
        r1, e1 = rc.view_factor_totals(scenes5[0], rpt, 7)
        t1 = tn = 1e30
        for _ in range(3):
            v0 = time.perf_counter(); rc.view_factor_totals(scenes5[0], rpt, 7); t1 = min(t1, time.perf_counter() - v0)
        dev_ms = 1e30
        for _ in range(3):
            v0 = time.perf_counter(); rg, eg = rc.view_factor_totals_multi(scenes5, rpt, 7); tn = min(tn, time.perf_counter() - v0)
            dev_ms = min(dev_ms, scenes5[0].last_kernel_ms())       # first launch -> reduced vectors on device 0, on device 0's clock (trace + reduce only)
        same = bool(np.array_equal(r1, rg) and np.array_equal(e1, eg)) and int(rg.sum()) == want
        md["view_factor_totals_c5"] = {"one_device_s": round(t1, 4), "all_devices_rays_s": round(tn, 4), "speedup": round(t1 / tn, 2), "same_vectors": same,
                                       "totals_ms": round(tn * 1e3, 3), "totals_device_ms": round(dev_ms, 3), "comm_init_ms": round(prep["comm_init_ms"], 2), "replicate_ms": round(replicate_ms, 2),
                                       "rccl_ranks": prep["rccl_ranks"], "reduce_bytes": 16 * n5,
                                       "checksum": zlib.crc32(rg.tobytes() + eg.tobytes()), "checksum_one_device": zlib.crc32(r1.tobytes() + e1.tobytes())}
        assert same, "view_factor_totals_multi differs from one device"
        # (1c) RAYS on the full matrix (the 10 GB ncclReduce, < 1x by DESIGN 5's own budget) only as a correctness run on distinct devices
        if forced <= 1:
            v0 = time.perf_counter(); rc.view_factors_multi(scenes5, rpt, 7, mode="rays", out=out_m); tr = time.perf_counter() - v0
            md["view_factors_c5_rays_rccl"] = {"seconds": round(tr, 4), "rccl_ranks": len(devices), "same_count": int(out_m.sum(dtype=np.int64)) == want}
            assert md["view_factors_c5_rays_rccl"]["same_count"], "RAYS (RCCL) counted different rays than one device"
        del out_m
        for t in scenes5:
            t.free()
        # (2) one host batch of closest_hit rays: contiguous shards, every device uploads / traces / downloads its own
        replicas = []
        for d in devices:
            t = rc.TLAS(d)
            for verts, meta in cfg["blas"]:
                t.add_geometry(verts, meta)
            for b, xf, ids in cfg["instances"]:
                t.push_instances(b, xf, ids)
            replicas.append(t.sync())
        big = np.concatenate([rays] * 4)                            # 16.8 M rays, 0.54 GB each way
        out_h = np.empty(len(big), dtype=rc.HIT_DT)
        replicas[0].trace(big, out=out_h)
        ref = out_h.copy()
        best1 = bestn = 1e30
        for _ in range(2):
            v0 = time.perf_counter(); replicas[0].trace(big, out=out_h); best1 = min(best1, time.perf_counter() - v0)
        for _ in range(2):
            v0 = time.perf_counter(); rc.trace_multi(replicas, big, out=out_h); bestn = min(bestn, time.perf_counter() - v0)
        md["host_batch_closest_hit"] = {"rays": len(big), "one_device_Mrays_s": round(len(big) / best1 / 1e6, 1), "all_devices_Mrays_s": round(len(big) / bestn / 1e6, 1),
                                        "speedup": round(best1 / bestn, 2), "identical_hits": bool(out_h.tobytes() == ref.tobytes()),
                                        "note": "host buffers in, host buffers out: 64 bytes per ray over PCIe, one link per device"}
        for t in replicas:
            t.free()
        extras["one_process_multi_device"] = md

    counts = load_json(COUNTS_FILE) or {}
    node_f = float(counts.get("node_fetches_per_ray", C3_NODE_FETCHES_PER_RAY))
    inst_f = float(counts.get("instance_entries_per_ray", C3_INST_ENTRIES_PER_RAY))
    counts_source = COUNTS_FILE if counts else "bench.py constants (round-1 oracle run)"
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # The oracle is the checker / CPU baseline only; nothing above this line touches it.
        from oracle import pyoracle as po
        o = po.Scene()
        for verts, meta in cfg["blas"]:
            o.add_blas(verts, meta)
        for b, xf, ids in cfg["instances"]:
            for x, i in zip(xf, ids):
                o.add_instance(b, x, int(i))
        o.build()
        # Harness: one worker per CPU this process may run on, pinned; the result array is allocated ONCE and its pages
        # are faulted in by the warm-up pass (a fresh 134 MB array per pass meant page faults under 256 threads inside the timed
        # region); the per-ray fetch counters are taken in ONE untimed instrumented pass, not in the timed ones.
        # The container may see every hardware thread of the host (sched_getaffinity: 256 on the GPU box) and still be limited to a CPU-time
        # quota by its cgroup (cpu.max = "1600000 100000" there: 16 CPUs' worth; with 256 runnable threads the kernel throttled the run
        # for 431 s of thread time and the rate FELL to 5 Mrays/s -- profiles/r03_cpu_scaling.txt).  One worker per CPU the quota pays for.
        po.pool_pin(True)
        visible = po.allowed_cpus()
        quota_cpus, quota_src = None, None
        for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            try:
                txt = open(path).read().split()
                if path.endswith("cpu.max"):
                    if txt[0] != "max":
                        quota_cpus, quota_src = int(txt[0]) / int(txt[1]), f"{path} = {' '.join(txt)}"
                else:
                    q = int(txt[0])
                    if q > 0:
                        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                        quota_cpus, quota_src = q / per, f"{path} = {q} / {per}"
                break
            except (OSError, ValueError, IndexError):
                continue
        cores = visible if not quota_cpus else max(1, min(visible, int(round(quota_cpus))))
        phys = set()
        cpu_model, pid, cid = "unknown", None, None
        try:
            for line in open("/proc/cpuinfo"):
                k, _, v = line.partition(":")
                k = k.strip()
                if k == "model name" and cpu_model == "unknown":
                    cpu_model = v.strip()
                elif k == "physical id":
                    pid = v.strip()
                elif k == "core id":
                    cid = v.strip()
                elif not line.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
        except OSError:
            pass
        physical_cores = min(len(phys), cores) if phys else cores  # cores the run can actually occupy: the quota's, when there is one
        ohits = np.zeros(n, dtype=rc.HIT_DT)
        o.trace(rays, nthreads=cores, out=ohits)  # warm: creates the worker pool, faults the result pages in
        passes = []
        for _ in range(5):  # 5 passes over the whole batch, best one reported (a single sub-second pass on 256 threads is noisy)
            c0 = time.perf_counter()
            o.trace(rays, nthreads=cores, out=ohits)
            passes.append(time.perf_counter() - c0)
        cdt = min(passes)
        _, cnt = o.trace(rays, nthreads=cores, counters=True)  # untimed: the reference algorithm's node / instance fetches per ray
        node_f, inst_f = float(cnt[:, 0].mean()), float(cnt[:, 1].mean())
        del cnt
        counts_source = "instrumented oracle, this run"
        same_canonical = bool(np.array_equal(ohits["primitive_id"], hits["primitive_id"]) and np.array_equal(ohits["instance_id"], hits["instance_id"])
                              and np.array_equal(ohits["t"].view(np.uint32), hits["t"].view(np.uint32)))
        fhits = o.trace(last_fresh_rays, nthreads=cores)   # the LAST TIMED STEP's batch (jittered rays) and what the timed loop left in its output buffer
        same_fresh = bool(np.array_equal(fhits["hit"], last_fresh_hits["hit"]) and np.array_equal(fhits["primitive_id"], last_fresh_hits["primitive_id"])
                          and np.array_equal(fhits["instance_id"], last_fresh_hits["instance_id"]) and np.array_equal(fhits["t"].view(np.uint32), last_fresh_hits["t"].view(np.uint32))
                          and np.array_equal(fhits["bary_u"].view(np.uint32), last_fresh_hits["bary_u"].view(np.uint32)) and np.array_equal(fhits["bary_v"].view(np.uint32), last_fresh_hits["bary_v"].view(np.uint32)))
        fresh_hit_frac = float(last_fresh_hits["hit"].mean())
        same = same_canonical and same_fresh
        # one thread on a bounded sample (every 16th ray: the same image, 262 144 rays, a fraction of a second) -- separates the
        # algorithm's per-core rate from the harness's scaling
        sample = np.ascontiguousarray(rays[::16])
        shits = np.zeros(len(sample), dtype=rc.HIT_DT)
        o.trace(sample, nthreads=1, out=shits)
        s0 = time.perf_counter()
        o.trace(sample, nthreads=1, out=shits)
        sdt = time.perf_counter() - s0
        single = len(sample) / sdt / 1e6
        cpu_baseline = {"value": round(n / cdt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "physical_cores": physical_cores, "hardware_threads_visible": visible,
                        "host_physical_cores": len(phys) or None, "cgroup_cpu_quota": quota_src or "none", "cpu_model": cpu_model, "kind": "port",
                        "sample": f"all {n} primary rays of the workload, closest_hit, C restatement of the reference algorithm "
                                  f"(oracle/, gcc -O2, persistent pool of {cores} pinned pthreads = min(sched_getaffinity count, cgroup CPU quota), dynamic 4096-ray chunks, result array "
                                  f"preallocated and pre-faulted, no instrumentation in the timed passes), best of 5 passes, {cdt:.3f} s per pass "
                                  f"(all: {', '.join(f'{p:.3f}' for p in passes)})",
                        "single_thread": {"value": round(single, 3), "unit": "Mrays/s", "cores": 1,
                                          "sample": f"every 16th ray of the batch ({len(sample)} rays), one pass, {sdt:.2f} s"},
                        "scaling_vs_physical_cores": round(n / cdt / 1e6 / (physical_cores * single), 3),
                        "gpu_matches_bit_exact": same,
                        "gpu_matches_bit_exact_what": f"the last timed step's output (its own jittered batch, {n} rays, hit fraction {fresh_hit_frac:.4f}: {same_fresh}) and the unjittered C3 batch the CPU passes were timed on ({same_canonical}); ids, t, u, v bits"}
        try:
            json.dump({"workload": "C3 primary rays, 2048 x 2048", "node_fetches_per_ray": node_f, "instance_entries_per_ray": inst_f,
                       "source": "oracle/ (instrumented reference algorithm), bench.py cpu_baseline leg"}, open(os.path.join(ROOT, COUNTS_FILE), "w"), indent=1)
        except OSError:
            pass

    if rank == 0:
        roofline = make_roofline(launch_ms, n, node_f, inst_f, counts_source, load_json(COUNTER_FILE) or {}, t.get_option("kernel"), args.res == 2048,
                                 mix=load_json(MIX_FILE), fingerprint=kernel_fingerprint())
        out = {
            "metric": "Mrays/s closest_hit (1M-tri TLAS)", "value": round(world * n * args.steps / elapsed / 1e6, 1), "unit": "Mrays/s",
            # `value` is the FIRST launch of a batch (every timed step traces rays never seen before).  Beside it, from the extras
            # of the same run (same scene, same ray count, 1 GPU): repeated_value = one buffer replayed with its learned claim order (rounds 1-5's
            # `value`); same_buffer_natural_order_value = one cache-warm buffer with the order switched off (rounds 4-5's first_touch_value);
            # moving_camera_value = a camera that moves every frame
            "repeated_value": None, "same_buffer_natural_order_value": None, "moving_camera_value": None,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3: TLAS of 256 rotated/scaled instances of one 4096-triangle BLAS (1 048 576 triangles), "
                                   f"{n} pinhole primary rays per GPU per step, closest_hit",
                       "scheduling": "first launch of the batch: every warm-up and timed step traces its OWN batch of rays (the C3 camera, each ray through a random point of its pixel; "
                                     f"{n_fresh} different 32 B/ray buffers resident in HBM), default options as shipped -- a batch the library has not seen runs in natural claim order "
                                     "and records nothing.  repeated_value: ONE buffer replayed, the order in which its 128-ray chunks are claimed learned from its earlier launches "
                                     "(option cost_order: the batch is recognised on the device by sample rays)",
                       "last_timed_step": order_state,
                       "entry_cull": "on (default): an instance whose conservative sphere the ray's segment misses is not entered -- the reference's traversal of it "
                                     "would test no triangle (DESIGN 4.1); every hit record identical with the option off (gpu_matches_bit_exact below is against the CPU oracle); "
                                     "the roofline's VALU counters are those of this kernel, the algorithmic bytes are the reference algorithm's",
                       "triangles": int(n_tris), "rays_per_step_per_gpu": n, "hit_fraction": round(hit_frac, 4),
                       "kernel": {-1: "auto (phased persistent, top level in LDS)", 0: "simple", 1: "persistent", 2: "voted", 3: "phased", 4: "phased + top level in LDS, 1024-thread workgroups", 5: "phased + top level in LDS", 6: "phased + tops of the TLAS / BLAS in LDS"}.get(t.get_option("kernel"), "option"), "parallelism": f"replicas x{world} (rays sharded, no collective)"},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "extras": extras,
        }
    else:
        out = None

    def emit():
        if out is None:
            return
        # RCCL prints its version banner through C stdio, which would otherwise be flushed at exit, AFTER the JSON line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        out["repeated_value"] = (extras.get("c3_repeated_batch") or {}).get("mrays_s")
        out["same_buffer_natural_order_value"] = (extras.get("c3_cost_order_off") or {}).get("mrays_s")
        out["moving_camera_value"] = (extras.get("c3_moving_camera") or {}).get("mrays_s")
        print(json.dumps(out), flush=True)  # the ONE JSON line, last thing on stdout

    if not args.no_extras and args.backend == "nccl":
        # The view-factor extra is the only measurement with collectives in it.  It runs first among the extras, with the headline
        # result already assembled, under a watchdog: if a rank fails inside a collective the others would wait for the process
        # group's time-out, and a hung extra must not cost the headline line -- rank 0 prints what it has, then every rank exits
        # NON-ZERO (a launcher must not read a hung run as a success).
        import threading
        finished = threading.Event()

        def watchdog():
            if not finished.wait(360.0):
                extras["view_factors_error"] = "timed out after 360 s (a rank left the collective sequence?)"
                emit()
                os._exit(3)

        if world > 1:
            threading.Thread(target=watchdog, daemon=True).start()
        guarded_extra("view_factors", extra_view_factors)
        finished.set()
    if use_dist:
        # every collective is behind us: the peers leave now instead of idling inside RCCL while rank 0 runs its single-GPU extras
        failed = torch.tensor([1 if any(k.endswith("_error") for k in extras) else 0], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            raise SystemExit(3 if int(failed.item()) else 0)
        if int(failed.item()) and "view_factors_error" not in extras:
            extras["view_factors_error"] = "a peer rank failed inside the view-factor extra"
    if not args.no_extras and rank == 0:
        guarded_extra("traces", extra_traces)
        guarded_extra("builds", extra_builds)
        guarded_extra("bvh4_collision", extra_bvh4_and_collision)
        if world == 1:
            guarded_extra("one_process_multi_device", extra_one_process_multi_device)
    emit()
    if any(k == "view_factors_error" for k in extras) and world > 1:
        raise SystemExit(3)


def dry_run(args, rank, world):
    """No GPU: the same control flow -- process group, barrier-bracketed timed region, max over ranks, peers leaving before rank 0's
    single-process tail, ONE JSON line from rank 0 -- with a sleep standing in for the launch.  Used by the CPU tests."""
    import torch
    import torch.distributed as dist
    use_dist = world > 1
    if use_dist:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    n = args.res * args.res
    for _ in range(args.warmup):
        time.sleep(0.001)
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "Mrays/s closest_hit (1M-tri TLAS)", "value": round(world * n * args.steps / elapsed / 1e6, 1), "unit": "Mrays/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "dry run (no GPU work)",
                          "config": {"workload": "dry run"}, "roofline": None, "cpu_baseline": None}), flush=True)
    return 0


if __name__ == "__main__":
    main()
