#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/mocap_full; mkdir -p $O; cd $ROOT
SMPLPP_DEBUG_SYNC=1 timeout -k 10 600 python tools/mocap_full.py 64 > $O/dbg_out.txt 2> $O/dbg_err.txt
tail -1 $O/dbg_out.txt
grep "project lists" $O/dbg_err.txt > $O/lists.txt
python3 - <<PY
import re, numpy as np
rows = [list(map(int, re.findall(r"(\d+)", l.split("project lists:")[1]))) for l in open("$O/lists.txt")]
a = np.array(rows)  # tasks, empty, overflow, nan, maxcnt
print("iterations", len(a), "tasks per iteration", a[0, 0])
print("empty lists: total %d, iterations with any %d, max per iteration %d" % (a[:, 1].sum(), (a[:, 1] > 0).sum(), a[:, 1].max()))
print("overflowing lists: total %d, iterations with any %d, max per iteration %d" % (a[:, 2].sum(), (a[:, 2] > 0).sum(), a[:, 2].max()))
print("nan: total %d" % a[:, 3].sum(), " max count: median %d, p90 %d, max %d" % (np.median(a[:, 4]), np.percentile(a[:, 4], 90), a[:, 4].max()))
for lo in range(0, len(a), 400):
    s = a[lo:lo + 400]
    print("  iterations %4d..: empty %5d overflow %5d maxcnt median %d" % (lo, s[:, 1].sum(), s[:, 2].sum(), np.median(s[:, 4])))
PY
rm -f $O/dbg_err.txt
