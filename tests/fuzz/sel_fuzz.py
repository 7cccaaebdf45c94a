import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import numpy as np, orc, gpu_api
from orc import Arr
o = orc.Api("orc"); g = gpu_api.GpuApi()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
TY = [orc.BOOL, orc.INT8, orc.UINT16, orc.INT32, orc.FLOAT, orc.INT64, orc.DOUBLE]
IT = [orc.INT8, orc.UINT8, orc.INT16, orc.UINT16, orc.INT32, orc.UINT32, orc.INT64, orc.UINT64]
bad = 0
def rep(tag, ok, t0):
    global bad
    bad += not ok
    print(tag, "ok" if ok else "MISMATCH", "%.1fs" % (time.time() - t0), flush=True)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    t = TY[int(rng.integers(0, len(TY)))]
    n = int(rng.choice([300_000, 2_000_000, 18_000_000]))
    v = orc.random_arr(rng, t, n, float(rng.choice([0, .01, .3])), int(rng.integers(0, 9)), null_count=int(rng.choice([-1, 0])) if False else None)
    sel = float(rng.choice([.01, .3, .5, .95]))
    f = orc.random_arr(rng, orc.BOOL, n, float(rng.choice([0, .02])), int(rng.integers(0, 9)), hi=sel)
    ns = int(rng.integers(0, 2))
    t0 = time.time()
    rep(("filter", t, n, sel, ns), g.filter(v, f, ns).same(o.filter(v, f, ns)), t0)
    m = int(rng.choice([1000, 1_200_000, 3_000_000]))
    itp = IT[int(rng.integers(0, len(IT)))]
    hi = min(n - 1, int(np.iinfo(orc.NP_OF[itp]).max))
    idx = orc.random_arr(rng, itp, m, float(rng.choice([0, .05])), int(rng.integers(0, 5)), lo=0, hi=hi)
    if rng.random() < .4:
        d = np.sort(idx.data); idx = Arr(itp, d, idx.valid, idx.offset, idx.length)
    t0 = time.time()
    rep(("take", t, n, itp, m), g.take(v, idx).same(o.take(v, idx)), t0)
# strings
for it in range(6):
    n = int(rng.choice([50_000, 400_000]))
    maxlen = int(rng.choice([3, 16, 60, 300]))
    lens = rng.integers(0, maxlen + 1, n + 3)
    items = [None if rng.random() < .05 else bytes(rng.integers(97, 123, int(l)).astype(np.uint8)) for l in lens]
    v = orc.BinArr(items, 3, n, pad=int(rng.integers(0, 9)))
    for sel in (.9, .4, .05):
        f = orc.random_arr(rng, orc.BOOL, n, .02, int(rng.integers(0, 9)), hi=sel)
        t0 = time.time()
        rep(("binary_filter", n, maxlen, sel), g.binary_filter(v, f, 0).same(o.binary_filter(v, f, 0)), t0)
    idx = orc.random_arr(rng, orc.INT32, 200_000, .05, 2, lo=0, hi=n - 1)
    loc = np.sort(idx.data) + rng.integers(-3, 4, len(idx.data)); loc = np.clip(loc, 0, n - 1)
    for ix in (idx, Arr(orc.INT32, loc, idx.valid, idx.offset, idx.length)):
        t0 = time.time()
        rep(("binary_take", n, maxlen), g.binary_take(v, ix).same(o.binary_take(v, ix)), t0)
print("bad:", bad)
