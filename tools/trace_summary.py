"""Condense a rocprofv3 --kernel-trace csv: per kernel name the durations (ms) of every dispatch, in launch order.
usage: python tools/trace_summary.py <dir-or-csv> [name-substring ...]"""
import csv, glob, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
want = sys.argv[2:]
out = {}
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if want and not any(w in n for w in want):
        continue
    out.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for n, v in out.items():
    print(f"{n:34s} " + " ".join(f"{x:8.2f}" for x in v[:12]))
