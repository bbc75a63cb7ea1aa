#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ts; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/trace_small.py
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"][:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "mdt" in r["Kernel_Name"] or "kp_setup" in r["Kernel_Name"] or "order_kernel" in r["Kernel_Name"]]
# the async phase: 100 requests after the first
import collections
seq = ks[2:2 + 200]
for i in range(0, 12):
    n, s, e = seq[i]
    gap = s - seq[i - 1][2] if i else 0
    print(f"{n:40s} dur {(e - s) / 1e3:8.1f} us   gap before {gap / 1e3:8.1f} us")
d = collections.defaultdict(list)
for i, (n, s, e) in enumerate(seq):
    d[n].append((e - s) / 1e3)
    if i: d["gap before " + n].append((s - seq[i - 1][2]) / 1e3)
for k, v in d.items():
    print(f"{k:60s} mean {sum(v) / len(v):8.1f} us  n={len(v)}")
PY
find $O -type f ! -name "*.txt" -delete
