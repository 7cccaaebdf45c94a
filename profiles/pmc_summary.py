#!/usr/bin/env python3
"""Per-kernel HBM bytes per dispatch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "at::native" in name or "rocclr" in name:
                continue
            acc[(name[:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, counter), v in acc.items():
        kb = sum(v) / len(v)
        if counter == "FETCH_SIZE":
            kb *= 2  # gfx950: FETCH_SIZE counts half of the streamed bytes
        print("%-72s %-11s avg %12.1f KB (n=%d)%s" % (name, counter, kb, len(v), "  [x2 applied]" if counter == "FETCH_SIZE" else ""))
