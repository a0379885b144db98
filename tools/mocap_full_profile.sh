#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/mocap_full; rm -rf $O; mkdir -p $O; cd $ROOT
timeout -k 10 300 python tools/mocap_full.py 64 2>/dev/null | tail -1 | tee $O/plain.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $ROOT/tools/mocap_full.py 64 > $O/traced.txt 2> $O/err.txt
tail -1 $O/traced.txt
python3 - <<PY
import csv, glob
f = glob.glob("$O/tr/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-56s calls %6s avg %9.2f us total %9.2f ms %6s%%" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
f = glob.glob("$O/tr/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# per-iteration period from the eval kernel starts in the second half of the run
ev = [int(r["Start_Timestamp"]) for r in rows if "ik_eval_kernel" in r["Kernel_Name"]]
import numpy as np
d = np.diff(np.array(ev[len(ev) // 2:])) / 1e3
print("eval-to-eval period: median %.1f us, mean %.1f, p90 %.1f, max %.1f (n=%d)" % (np.median(d), d.mean(), np.percentile(d, 90), d.max(), len(d)))
for name in ("ik_eval_kernel", "ik_solve_kernel", "proj_scan_kernel", "proj_finish_kernel"):
    v = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]])
    v = v[len(v) // 2:]
    print("%-20s median %.1f mean %.1f p90 %.1f max %.1f" % (name, np.median(v), v.mean(), np.percentile(v, 90), v.max()))
PY
rm -rf $O/tr
