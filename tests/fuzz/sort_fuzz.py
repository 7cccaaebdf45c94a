import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import numpy as np, orc, gpu_api
from orc import Arr
o = orc.Api("orc"); g = gpu_api.GpuApi()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
T = {"i32": orc.INT32, "u32": orc.UINT32, "i64": orc.INT64, "u64": orc.UINT64, "f32": orc.FLOAT, "f64": orc.DOUBLE}
def gen(tn, dist, n):
    isf = tn[0] == "f"
    info = None if isf else np.iinfo(orc.NP_OF[T[tn]])
    if dist == "full":
        return rng.standard_normal(n) * 1e30 if isf else rng.integers(info.min, info.max, n, dtype=orc.NP_OF[T[tn]], endpoint=True)
    if dist == "small":
        return rng.random(n) * 2 - 1 if isf else rng.integers(max(info.min, -500), 500, n).astype(orc.NP_OF[T[tn]])
    if dist == "clusters":
        c = np.array([-1e-20, 1e-20, 5e10, -5e10]) if isf else np.array([max(info.min, -2**30) + 7, 0, 3, min(info.max, 2**31 - 9) - 9]).astype(np.float64)
        x = c[rng.integers(0, 4, n)]
        return x * (1 + rng.random(n) * 1e-6) if isf else (x + rng.integers(0, 5000, n)).astype(orc.NP_OF[T[tn]])
    if dist == "magnitudes":
        m = 10.0 ** rng.uniform(-30, 30, n) * np.where(rng.random(n) < .5, -1, 1)
        return m if isf else np.clip(m, float(info.min) / 2, float(info.max) / 2).astype(orc.NP_OF[T[tn]])
    if dist == "sorted":
        x = np.sort(gen(tn, "full", n)); return x
    if dist == "reversed":
        return np.sort(gen(tn, "full", n))[::-1].copy()
    if dist == "dups":
        pool = gen(tn, "full", 1000); return pool[rng.integers(0, 1000, n)]
    if dist == "nearconst":
        x = np.full(n, 42, orc.NP_OF[T[tn]]); x[rng.integers(0, n, 100)] = gen(tn, "full", 100); return x
bad = 0
for tn in T:
    for dist in ("full", "small", "clusters", "magnitudes", "sorted", "reversed", "dups", "nearconst"):
        n = int(rng.integers(2_100_000, 3_200_000))
        d = gen(tn, dist, n).astype(orc.NP_OF[T[tn]])
        pn = float(rng.choice([0, .01, .4]))
        if tn[0] == "f" and rng.random() < .5: d[rng.random(n) < .002] = np.nan
        valid = (rng.random(n) >= pn) if pn else None
        order = int(rng.integers(0, 2))
        nch = int(rng.choice([1, 1, 3]))
        if nch == 1:
            a = Arr(T[tn], d, valid)
            t0 = time.time(); w = o.sort_indices(a, order); r = g.sort_indices(a, order)
        else:
            cuts = [0, n // 3, n // 3 + 5, n]
            chunks = [Arr(T[tn], d[cuts[i]:cuts[i + 1]], None if valid is None else valid[cuts[i]:cuts[i + 1]]) for i in range(3)]
            t0 = time.time(); w = o.sort_indices(chunks, order); r = g.sort_indices(chunks, order)
        ok = np.array_equal(w, r)
        bad += not ok
        print(tn, dist, n, "nulls", pn, "order", order, "chunks", nch, "ok" if ok else "MISMATCH %d" % int((w != r).sum()), "%.1fs" % (time.time() - t0), flush=True)
print("bad:", bad)
