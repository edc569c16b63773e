#!/bin/bash
# BVH4 vs BVH2 on the same single-BLAS scene (C2) under the same counters: is a 4-wide BLAS traversal bound differently?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02bvh4; mkdir -p $O
python3 tools/bvh4_probe.py > $O/probe.txt 2>&1; cat $O/probe.txt | tail -8
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TD_TD_BUSY_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/set$i -- python3 tools/bvh4_probe.py > $O/set$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r02bvh4/set*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "bvh4 k_trace4<closest>" if "k_trace4<false" in r["Kernel_Name"] else ("bvh2 kernel5<closest>" if "k_trace_phased_lds<false" in r["Kernel_Name"] else None)
        if k and int(r["Grid_Size"]) > 100000:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/r02bvh4/set1/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "bvh4 k_trace4<closest>" if "k_trace4<false" in r["Kernel_Name"] else ("bvh2 kernel5<closest>" if "k_trace_phased_lds<false" in r["Kernel_Name"] else None)
        if k:
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, c in agg.items():
    # dispatches alternate between the coherent and the incoherent 1 M-ray batch (20 + 20 each); report the coherent half = first 20 big launches after the warm-up
    out[k] = {n: sum(v) / len(v) for n, v in c.items()}
    out[k]["mean_ns"] = sum(dur[k]) / max(len(dur[k]), 1)
    out[k]["launches"] = len(dur[k])
json.dump(out, open("gpurun_out/r02bvh4/pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
