#!/usr/bin/env python3
"""Per-kernel HBM bytes from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs: average per dispatch,
and the total per bench step (all dispatches of the run / number of steps; a kernel that is
launched twice per step with one launch gated off on the device must not halve the figure).
The number of steps is the dispatch count of the op's first kernel of every step."""
import collections, csv, glob, sys

STEP_MARKERS = ("amdarrow::k_prepare", "amdarrow::k_filter_tile_counts", "amdarrow::k_sort_count", "amdarrow::k_validity",
                "amdarrow::k_boolean", "amdarrow::k_aggregate<", "amdarrow::k_take", "amdarrow::k_rawc_scan<unsigned long, 1>", "amdarrow::k_sort_hist")
for d in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "at::native" in name or "rocclr" in name:
                continue
            acc[(name[:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    steps, counter, grand = 0, None, 0.0
    for (name, c), v in acc.items():
        counter = c
        scale = 2.0 if c == "FETCH_SIZE" else 1.0  # gfx950: FETCH_SIZE counts half of the streamed bytes
        kb = sum(v) * scale
        grand += kb
        if any(m in name for m in STEP_MARKERS):
            steps = max(steps, len(v))
        print("%-72s %-11s avg %12.1f KB (n=%d)%s" % (name, c, kb / len(v), len(v), "  [x2 applied]" if scale == 2.0 else ""))
    if counter and steps:
        print("PER_STEP_TOTAL %s %.1f KB over %d steps" % (counter, grand / steps, steps))
        # a steady-state step: the kernels that run in at least half of the steps, each with its average bytes (the
        # group-by's first call of a shape runs other kernels -- it learns the hint the later calls start from)
        steady = 0.0
        for (name, c), v in acc.items():
            if 2 * len(v) >= steps:
                steady += sum(v) * (2.0 if c == "FETCH_SIZE" else 1.0) / len(v) * max(1, round(len(v) / steps))
        print("PER_STEP_STEADY %s %.1f KB (kernels present in >= half of the %d steps)" % (counter, steady, steps))
