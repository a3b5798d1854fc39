"""Condenses a profiles/run_profiles.sh output directory (gpurun_out/prof_<tag>) into the small CSVs
kept under profiles/<tag>/: kernel_stats.csv (rocprofv3 --kernel-trace --stats) and pmc_summary.csv
(mean counter value per launch and kernel).  With --latest <workload> it also writes
profiles/pmc_latest.json, which bench.py reads for roofline.traffic.
usage: python profiles/summarise.py <tag> [--latest <workload>]"""
import collections
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)

def newest(pattern):
    """gpurun merges a run's files into what earlier runs left: of several runs' files only the latest counts."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


rows = []
for f in newest(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    rows += list(csv.DictReader(open(f)))
with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "calls", "total_ns", "average_ns", "percentage", "min_ns", "max_ns"])
    for r in rows:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

acc = collections.defaultdict(list)
for f in [g for d in glob.glob(os.path.join(src, "pmc_*")) if os.path.isdir(d) for g in newest(os.path.join(d, "*", "*counter_collection.csv"))]:
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(os.path.join(dst, "pmc_summary.csv"), "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "counter", "launches", "mean_per_launch"])
    for (k, c), v in sorted(acc.items()):
        w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])
print(open(os.path.join(dst, "kernel_stats.csv")).read())
print(open(os.path.join(dst, "pmc_summary.csv")).read())

if "--latest" in sys.argv:
    workload = sys.argv[sys.argv.index("--latest") + 1]
    def mean(counter, frag):
        # the trace kernel has a write-through twin (asynchronous renders: bench.py's end_to_end leg; last template argument 1, "true" in
        # older builds); the timed steps run the plain one
        v = [x for (k, c), xs in acc.items() if c == counter and frag in k and "true>" not in k and ", 1>(" not in k for x in xs]
        return sum(v) / len(v) if v else 0.0
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so reads are doubled
    trace = 2 * 1024 * mean("FETCH_SIZE", "pt_k_trace") + 1024 * mean("WRITE_SIZE", "pt_k_trace")
    resolve = 2 * 1024 * mean("FETCH_SIZE", "pt_k_resolve") + 1024 * mean("WRITE_SIZE", "pt_k_resolve")
    # one file for every profiled workload: the headline one at the top level (as before), the others under "also"
    path = os.path.join(root, "profiles", "pmc_latest.json")
    try:
        latest = json.load(open(path))
    except Exception:
        latest = {}
    entry = {"workload": workload, "n_gpus": 1, "hbm_bytes_per_launch": int(trace),
             "hbm_bytes_resolve_kernel": int(resolve),
             "source": f"profiles/{tag}/pmc_summary.csv: pt_k_trace, 2 x FETCH_SIZE + WRITE_SIZE (KiB), separate --pmc passes"}
    if "--scattered" in sys.argv:
        # 64-byte requests of a scattered read pattern are tallied at their size: no doubling.  The counter sits on the L2's
        # memory side, so lines the Infinity Cache serves are in it: an upper bound of the HBM bytes.
        entry["hbm_bytes_per_launch"] = int(1024 * mean("FETCH_SIZE", "pt_k_trace") + 1024 * mean("WRITE_SIZE", "pt_k_trace"))
        entry["source"] = f"profiles/{tag}/pmc_summary.csv: pt_k_trace, FETCH_SIZE + WRITE_SIZE (KiB; scattered 64-byte reads, Infinity Cache hits included: an upper bound), separate --pmc passes"
    if workload == "demo-1080p-64spp-d8" or not latest:
        also = latest.get("also", {})
        latest = entry
        if also:
            latest["also"] = also
    else:
        latest.setdefault("also", {})[workload] = entry
    json.dump(latest, open(path, "w"), indent=1)
    print(open(path).read())
