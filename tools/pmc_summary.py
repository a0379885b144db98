"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean of each counter per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "skin_kernel" not in k and "pose_kernel" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 3:]  # skip warm-up dispatches
        print("   %-28s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
