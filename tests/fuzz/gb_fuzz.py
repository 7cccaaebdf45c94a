import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import numpy as np, orc, gpu_api
from orc import Arr
o = orc.Api("orc")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
KT = {"i8": orc.INT8, "u16": orc.UINT16, "i32": orc.INT32, "u32": orc.UINT32, "i64": orc.INT64, "u64": orc.UINT64, "f32": orc.FLOAT, "f64": orc.DOUBLE}
VT = {"i64": orc.INT64, "f64": orc.DOUBLE, "i32": orc.INT32, "u8": orc.UINT8, "f32": orc.FLOAT}
def keys(tn, dist, n):
    np_t = orc.NP_OF[KT[tn]]
    isf = tn[0] == "f"
    if tn == "i8": base = rng.integers(-100, 100, n)
    elif dist == "few": base = rng.integers(0, 300, n)
    elif dist == "1m": base = rng.integers(0, 1_000_000, n)
    elif dist == "3m": base = rng.integers(0, 3_000_000, n)
    elif dist == "sorted": base = np.sort(rng.integers(0, 200_000, n))
    elif dist == "zipf": base = np.minimum(rng.zipf(1.3, n), 500_000)
    elif dist == "hot": base = np.where(rng.random(n) < .3, 7, rng.integers(0, 400_000, n))
    elif dist == "wide":
        if tn in ("i64", "u64"):
            pool = rng.integers(0, 2**63 - 1, 150_000); base = pool[rng.integers(0, len(pool), n)]
        else: base = rng.integers(0, 60_000, n) * 7919
    elif dist == "offset": base = rng.integers(0, 50_000, n) + (10**12 if tn in ("i64", "u64") else 20_000)
    if tn == "u16": base = base % 65536
    if isf: return (base.astype(np.float64) * 0.5 - 1000).astype(np_t)
    return base.astype(np_t)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    gpu = gpu_api.GpuApi() if it % 5 == 0 else gpu
    tn = str(rng.choice(list(KT))); dist = str(rng.choice(["few", "1m", "3m", "sorted", "zipf", "hot", "wide", "offset"]))
    vn = str(rng.choice(list(VT)))
    n = int(rng.integers(300_000, 2_500_000))
    koff, voff, moff = int(rng.integers(0, 9)), int(rng.integers(0, 9)), int(rng.integers(0, 9))
    kd = keys(tn, dist, n + koff)
    kp = float(rng.choice([0, 0, .02, .3]))
    k = Arr(KT[tn], kd, (rng.random(n + koff) >= kp) if kp else None, koff, n)
    v = orc.random_arr(rng, VT[vn], n, float(rng.choice([0, .05])), voff, lo=-1000, hi=1000)
    sel = float(rng.choice([1.0, .5, .05, .9]))
    m = orc.random_arr(rng, orc.BOOL, n, float(rng.choice([0, .03])), moff, hi=sel)
    t0 = time.time()
    want = o.filter_groupby_sum(v, k, m)
    reps = 2 if rng.random() < .5 else 1
    ok = True
    for rep in range(reps):
        got = gpu.filter_groupby_sum(v, k, m)
        for x, y in zip(want, got):
            if x.dtype == np.float64:
                same = x.shape == y.shape and np.allclose(np.nan_to_num(x), np.nan_to_num(y), rtol=1e-9, atol=1e-6)
            else:
                same = np.array_equal(x, y)
            ok = ok and same
    bad += not ok
    print(it, tn, dist, vn, n, "knull", kp, "sel", sel, "offs", koff, voff, moff, "groups", len(want[0]), "ok" if ok else "MISMATCH", "%.1fs" % (time.time() - t0), flush=True)
print("bad:", bad)
