#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02calib; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib || exit 1
/tmp/fetch_calib > $O/plain.txt 2>&1; cat $O/plain.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- /tmp/fetch_calib > $O/fetch.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/rdreq -- /tmp/fetch_calib > $O/rdreq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/l2 -- /tmp/fetch_calib > $O/l2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("fetch", "rdreq", "l2"):
    for f in glob.glob(f"gpurun_out/r02calib/{d}/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:24], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(d, k, [round(x) for x in v])
print("requested: k_gather64 %d bytes per launch, k_stream %d bytes per launch; FETCH_SIZE is in KiB" % (256 * 8 * 4 * 256 * 16 * 64, 4 << 30))
PY
