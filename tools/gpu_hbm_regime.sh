#!/bin/bash
# HBM-bound regime: physical HBM bytes (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes) vs algorithmic bytes.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02hbm; mkdir -p $O
python3 tools/hbm_regime.py --out $O/regime.json > $O/regime.log 2>&1; cat $O/regime.log | tail -4
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-20)
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$tag -- python3 tools/hbm_regime.py --no-stats > $O/pmc_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
O = "gpurun_out/r02hbm"
reg = json.load(open(f"{O}/regime.json"))
# per scene: the 5 timed launches of the default kernel are consecutive k_trace dispatches; scenes come in the order of regime.json
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/pmc_*/**/*_counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_trace" in r["Kernel_Name"]]
    by_counter = collections.defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, vals in by_counter.items():
        n_scene = len(reg)
        k = len(vals) // n_scene
        for i, key in enumerate(reg):
            per[key][name] = vals[i * k:(i + 1) * k]
out = {"method": "rocprofv3 --pmc <counter> --kernel-trace, one pass per counter; FETCH_SIZE x2 (gfx950 reports half of a 16 B/lane coalesced stream; the node gathers here are 16-byte loads too), WRITE_SIZE as reported, KiB; amplification = physical HBM bytes / algorithmic bytes of the reference algorithm",
       "hbm_peak_GBs": 8000.0, "scenes": {}}
for key, e in reg.items():
    c = {k: sum(v) / len(v) for k, v in per[key].items() if v}
    phys = c.get("FETCH_SIZE", 0) * 2048 + c.get("WRITE_SIZE", 0) * 1024
    t = e["ms"] * 1e-3
    e2 = dict(e)
    e2.update({"counters_mean_per_launch": c, "hbm_physical_bytes_per_launch": phys, "hbm_physical_GBs": phys / t / 1e9, "hbm_physical_frac": phys / t / 1e9 / 8000.0,
               "algorithmic_GBs": e["algorithmic_bytes_per_launch"] / t / 1e9, "fetch_amplification": phys / e["algorithmic_bytes_per_launch"],
               "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)})
    out["scenes"][key] = e2
json.dump(out, open(f"{O}/r02_hbm_regime.json", "w"), indent=1)
print(json.dumps({k: {kk: v[kk] for kk in ("mrays_s", "hbm_physical_GBs", "hbm_physical_frac", "algorithmic_GBs", "fetch_amplification", "l2_hit_rate")} for k, v in out["scenes"].items()}, indent=1))
PY
