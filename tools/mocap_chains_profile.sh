#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/mocap_chains; rm -rf $O; mkdir -p $O; cd $ROOT
# usage (GPU box, repo root): bash tools/mocap_chains_profile.sh [chains, default 8] [latent]
R=${1:-8}; L=${2:-}
timeout -k 10 300 python tools/mocap_full.py $R 800 $L 2>/dev/null | tail -1 | tee $O/plain.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $ROOT/tools/mocap_full.py $R 800 $L > $O/traced.txt 2> $O/err.txt
python3 - <<PY
import csv, glob, numpy as np
f = glob.glob("$O/tr/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [int(r["Start_Timestamp"]) for r in rows if "ik_eval_kernel" in r["Kernel_Name"]]
d = np.diff(np.array(ev[len(ev) // 2:])) / 1e3
print("eval-to-eval period: median %.1f us, mean %.1f" % (np.median(d), d.mean()))
for name in ("ik_eval_kernel", "ik_solve_kernel", "vposer_jac2_kernel<1, true>", "vposer_jac2_kernel<1, false>", "proj_scan_kernel", "proj_finish_kernel", "pose_kernel", "skin_kernel", "ik_seq_frame", "streamOpsWait"):
    if not any(name in r["Kernel_Name"] for r in rows): continue
    v = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]])
    v = v[len(v) // 2:]
    print("%-30s median %.1f mean %.1f p90 %.1f" % (name, np.median(v), v.mean(), np.percentile(v, 90)))
tail = rows[-40:-14]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-44s start %9.2f us  end %9.2f us  dur %7.2f" % (r["Kernel_Name"][:44], s / 1e3, e / 1e3, (e - s) / 1e3))
PY
rm -rf $O/tr
