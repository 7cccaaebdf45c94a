import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import numpy as np, orc, gpu_api
from orc import Arr
o = orc.Api("orc"); g = gpu_api.GpuApi()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
TY = [orc.INT8, orc.UINT8, orc.INT16, orc.UINT16, orc.INT32, orc.UINT32, orc.INT64, orc.UINT64, orc.FLOAT, orc.DOUBLE]
def col(t, n):
    npt = orc.NP_OF[t]
    kind = str(rng.choice(["few", "mid", "full", "clusters"]))
    if t in (orc.FLOAT, orc.DOUBLE):
        if kind == "few": d = rng.integers(-3, 4, n) * 0.5
        elif kind == "mid": d = rng.random(n) * 100 - 30
        elif kind == "full": d = rng.standard_normal(n) * 1e20
        else: d = np.where(rng.random(n) < .5, -1e-10, 1e10) * (1 + rng.integers(0, 50, n) * 1e-3)
        d = d.astype(npt)
        if rng.random() < .5: d[rng.random(n) < .01] = np.nan
        if rng.random() < .5: d[rng.random(n) < .01] = -0.0
    else:
        info = np.iinfo(npt)
        if kind == "few": d = rng.integers(max(info.min, -3), 4, n)
        elif kind == "mid": d = rng.integers(max(info.min, -30000), min(info.max, 30000), n)
        elif kind == "full": d = rng.integers(info.min, info.max, n, dtype=npt, endpoint=True)
        else: d = np.where(rng.random(n) < .5, info.min + 5, info.max - 100).astype(np.float64) + rng.integers(0, 50, n)
        d = np.asarray(d).astype(npt)
    p = float(rng.choice([0, 0, .02, .3]))
    return Arr(t, d, (rng.random(n) >= p) if p else None), kind
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    n = int(rng.choice([150_000, 4_300_000, 4_300_000]))
    nk = int(rng.integers(2, 4))
    cols = [col(TY[int(rng.integers(0, len(TY)))], n) for _ in range(nk)]
    orders = [int(rng.integers(0, 2)) for _ in range(nk)]
    keys = [c[0] for c in cols]
    t0 = time.time()
    w = o.sort_indices_multi(keys, orders); r = g.sort_indices_multi(keys, orders)
    ok = np.array_equal(w, r)
    bad += not ok
    print(it, n, [(c[0].type_id, c[1], c[0].valid is not None) for c in cols], orders, "ok" if ok else "MISMATCH %d" % int((w != r).sum()), "%.1fs" % (time.time() - t0), flush=True)
print("bad:", bad)
