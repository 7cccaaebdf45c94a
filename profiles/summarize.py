#!/usr/bin/env python3
"""Print the per-kernel table of a rocprofv3 kernel_stats.csv (our kernels only)."""
import csv, glob, sys
for pat in sys.argv[1:]:
    for f in glob.glob(pat):
        print("==", f)
        for r in csv.DictReader(open(f)):
            if "at::native" in r["Name"] or "rocclr" in r["Name"]:
                continue
            print("  %-78s calls %4s avg %9.1f us  %6.2f%%" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
