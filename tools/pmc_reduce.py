"""Reduce rocprofv3 --pmc CSV output (one row per dispatch and counter) to per-kernel averages."""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]


def short(k):
    m = re.search(r"(\w*(?:dcn|conv)\w*(?:<[^>]*>)?)", k)
    return m.group(1) if m else k[:70]

acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "dcn" not in k and "conv" not in k:
            continue
        acc[short(k)][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "dcn" in k or "conv" in k:
            dur[short(k)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    d = dur.get(k, [])
    print("== %s  launches/pass=%d  avg_us(profiled)=%.1f" % (k, len(d) // max(1, len(glob.glob(os.path.join(root, 'p*.log')))),
                                                              sum(d) / max(1, len(d))))
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-32s avg/launch %.6g   (n=%d)" % (c, sum(v) / len(v), len(v)))
