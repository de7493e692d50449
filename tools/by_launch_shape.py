"""Reduce a rocprofv3 kernel trace (…_kernel_trace.csv) per kernel AND launch shape (grid size): count, average,
minimum and maximum duration -- the stats CSV of rocprofv3 merges every shape of a kernel into one row.
  python tools/by_launch_shape.py <kernel_trace.csv> [substring ...] > profiles/<name>_by_launch_shape.csv"""
import csv, re, sys
from collections import defaultdict

rows = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
        continue
    grid = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = "x".join(r.get(k, "?") for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
    rows[(re.sub(r"\s+", " ", name)[:160], grid, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_threads", "workgroup", "launches", "avg_us", "min_us", "max_us", "total_us"])
for (k, g, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([k, g, wg, len(v), "%.2f" % (sum(v) / len(v)), "%.2f" % min(v), "%.2f" % max(v), "%.1f" % sum(v)])
