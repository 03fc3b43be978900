"""Timeline of the ycnr kernels of the LAST iteration in a rocprofv3 --kernel-trace CSV (kernel_trace.csv): start offset, duration,
queue, grid.  usage: python3 profiles/trace_timeline.py <dir> [n_last_kernels]"""
import csv, glob, sys

root = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "ycnr" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n:]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f us  +%8.1f us  q%-3s grid %-9s wg %-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")),
                                                                   r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), r["Kernel_Name"].replace("ycnr::", "")[:70]))
