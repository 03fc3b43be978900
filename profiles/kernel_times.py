"""Prints the ycnr kernels of a rocprofv3 --kernel-trace --stats CSV: name, calls, average ns."""
import csv, glob, sys

root = sys.argv[1] if len(sys.argv) > 1 else "."
for f in glob.glob(root + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ycnr" in r["Name"]:
            print("%-72s %4s %12.0f" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])))
