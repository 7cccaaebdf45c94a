#!/usr/bin/env python3
"""Speed of the real reference build (oracle/_ref) next to the C restatement (oracle/oracle.c) on the
same inputs, 1 thread, this container.  Fills the table of BASELINE.md section 5.  Test infrastructure."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import orc  # noqa: E402

rng = np.random.default_rng(0x0ff1ce)
n = 4_000_000
ns = 1_000_000


def nullable(t, data, p):
    return orc.Arr(t, data, rng.random(len(data)) >= p)


vals = nullable(orc.INT64, rng.integers(-100, 101, n), .01)
mask = orc.Arr(orc.BOOL, rng.random(n) < .5)
keys = orc.Arr(orc.INT32, rng.integers(0, 1_000_000, n))
fvals = nullable(orc.DOUBLE, rng.random(n) * 1e3, .01)
ones = orc.Arr(orc.BOOL, np.ones(n, bool))
a, b = orc.Arr(orc.INT64, rng.integers(-100, 101, n)), orc.Arr(orc.INT64, rng.integers(-100, 101, n))
su = orc.Arr(orc.UINT64, rng.integers(0, 2**64, ns, dtype=np.uint64))
cases = [
    ("filter int64+validity, 50 %", n, lambda api: api.filter(vals, mask)),
    ("take int64 by monotonic uint32", n // 2, None),
    ("sort_indices uint64", ns, lambda api: api.sort_indices([su])),
    ("Filter->GroupBy{sum,count} int64, 1 M keys", n, lambda api: api.filter_groupby_sum(vals, keys, mask)),
    ("GroupBy{sum,count} float64, 1 M keys", n, lambda api: api.filter_groupby_sum(fvals, keys, ones)),
    ("greater(int64,int64)", n, lambda api: api.compare(orc.GT, a, b)),
    ("add(int64,int64)", n, lambda api: api.arith(0, a, b)),
]
apis = {"ref": orc.Api("ref"), "orc": orc.Api("orc")}
idx = orc.Arr(orc.UINT32, apis["orc"].take_indices(mask).values)
cases[1] = (cases[1][0], idx.length, lambda api: api.take(vals, idx))
print("| op | reference rows/s (container) | repo CPU restatement rows/s (container) | ratio |\n|---|---|---|---|")
for name, rows, fn in cases:
    r = {}
    for k, api in apis.items():
        fn(api)
        t0 = time.time(); reps = 0
        while time.time() - t0 < 3.0:
            fn(api); reps += 1
        r[k] = rows * reps / (time.time() - t0)
    print("| %s | %.1f M | %.1f M | %.2f |" % (name, r["ref"] / 1e6, r["orc"] / 1e6, r["orc"] / r["ref"]))
