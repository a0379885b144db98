"""profiles/traffic.json from a PMC summary (tools/pmc_summary.py output of the separate --pmc passes of
tools/collect_profiles.sh): HBM bytes per launch of the fused FK kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 — FETCH_SIZE
and WRITE_SIZE are in KB, and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming reads
(MI355X_MICROARCH.md, HBM section; this kernel's reads are 16 B-per-lane LDS-DMA and register loads).
usage: python tools/make_traffic_json.py <pmc_summary.txt> <out.json> [frames] [tag] [kernel_stats.csv ...]
(tag + kernel_stats.csv: the round tag of the collection and the kernel-trace stats of THE SAME collection run — the fused kernel's
average duration under rocprofv3 is recorded beside the byte counts, so the reader of bench.py's `roofline.traffic` can see which
run the constant came from: bench.py copies `tag` into `roofline.traffic_source`)"""
import json, re, sys

src, dst = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
tag = sys.argv[4] if len(sys.argv) > 4 else None
stats = sys.argv[5:]  # kernel-trace stats of the same collection (one file per form traced)
cur, vals = None, {}
for ln in open(src):
    if not ln.startswith(" "):
        cur = ln.strip()
        vals.setdefault(cur, {})
        continue
    m = re.match(r"\s+(\S+)\s+mean\s+(\S+)", ln)
    if m and cur:
        vals[cur][m.group(1)] = float(m.group(2))
out = {}
for k, c in vals.items():
    m = re.search(r"skin_kernel_?([a-z])?", k)
    if not m or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    form = m.group(1) or "v"
    hbm = int(round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024))
    out["skin_kernel_%s_hbm_bytes_per_launch_n%d" % (form, n)] = hbm
    out["skin_kernel_%s_detail" % form] = {
        "kernel": k, "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"], "algorithmic_bytes": 19347120 + 83020 * n,
        "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES"), "wave_quad_cycles": c.get("SQ_WAVE_CYCLES"),
        "mfma_busy_of_wave_cycles": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_WAVE_CYCLES"])) if c.get("SQ_WAVE_CYCLES") else None,
        "l2_hit": c.get("TCC_HIT_sum"), "l2_miss": c.get("TCC_MISS_sum"),
    }
out["how"] = ("rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (tools/collect_profiles.sh), mean per launch after warm-up; "
              "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md")
if tag:
    out["tag"] = tag
import csv

for st in stats:
    for r in csv.DictReader(open(st)):
        m = re.search(r"skin_kernel_?([a-z])?", r["Name"])
        if m:
            out["skin_kernel_%s_rocprofv3_avg_us_same_run" % (m.group(1) or "v")] = float(r["AverageNs"]) / 1e3
            out["skin_kernel_%s_rocprofv3_calls_same_run" % (m.group(1) or "v")] = int(r["Calls"])
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
