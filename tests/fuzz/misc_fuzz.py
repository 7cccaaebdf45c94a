import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import numpy as np, orc, gpu_api
from orc import Arr
o = orc.Api("orc"); g = gpu_api.GpuApi()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
KT = [orc.INT8, orc.UINT16, orc.INT32, orc.UINT32, orc.INT64, orc.UINT64, orc.FLOAT, orc.DOUBLE, orc.BOOL]
VT = [orc.INT64, orc.DOUBLE, orc.INT32, orc.UINT8, orc.FLOAT, orc.UINT64]
bad = 0
def rep(tag, ok, t0):
    global bad
    bad += not ok
    print(tag, "ok" if ok else "MISMATCH", "%.1fs" % (time.time() - t0), flush=True)
def close(a, b):
    if a.dtype.kind == "f": return a.shape == b.shape and np.allclose(np.nan_to_num(a), np.nan_to_num(b), rtol=1e-9, atol=1e-6)
    return np.array_equal(a, b)
def keycol(t, n, ng, off):
    if t == orc.BOOL: return Arr(t, rng.random(n + off) < .5, (rng.random(n + off) > .1) if rng.random() < .5 else None, off, n)
    base = rng.integers(0, ng, n + off)
    if t == orc.INT8: base = base % 200 - 100
    if t == orc.UINT16: base = base % 65536
    if t in (orc.INT64, orc.UINT64) and rng.random() < .5: base = base * 2**40 + 12345
    d = (base * 0.25 - 7).astype(orc.NP_OF[t]) if t in (orc.FLOAT, orc.DOUBLE) else base.astype(orc.NP_OF[t])
    p = float(rng.choice([0, 0, .05]))
    return Arr(t, d, (rng.random(n + off) >= p) if p else None, off, n)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    n = int(rng.integers(300_000, 1_500_000))
    # grouper over 1-3 key columns in 1-3 batches
    nk = int(rng.integers(1, 4))
    ng = int(rng.choice([50, 5000, 400_000]))
    ts = [KT[int(rng.integers(0, len(KT)))] for _ in range(nk)]
    nb = int(rng.integers(1, 4))
    batches = []
    for b in range(nb):
        m = n if b == 0 else int(rng.integers(1000, 400_000))
        batches.append([keycol(t, m, ng, int(rng.integers(0, 5))) for t in ts])
    t0 = time.time()
    def same_groups(w, r):
        (io, ngo, uo), (ig, ngg, ug) = w, r
        if ngo != ngg or not all(np.array_equal(x, y) for x, y in zip(io, ig)):
            print("  ids / group count differ:", ngo, ngg, [bool(np.array_equal(x, y)) for x, y in zip(io, ig)])
            return False
        for c, ((va, vb, nca), (vc, vd, ncb)) in enumerate(zip(uo, ug)):
            ok_v = np.array_equal(vb, vd) and nca == ncb
            m = vb.astype(bool)[:len(va)] if len(vb) >= len(va) else None
            ok_d = ok_v and (np.array_equal(va[m], vc[m], equal_nan=va.dtype.kind == "f") if m is not None else np.array_equal(va, vc))
            if not ok_d:
                print("  uniques of column", c, "differ: validity", ok_v, "null counts", nca, ncb)
                return False
        return True
    try:
        w = o.group_ids(batches)
        r = g.group_ids(batches)
        ok = same_groups(w, r)
    except orc.StatusError as e:
        try:
            g.group_ids(batches); ok = False
        except orc.StatusError as e2:
            ok = e.code == e2.code
    rep(("group_ids", ts, ng, nb, n), ok, t0)
    # hash aggregates over ids
    ngr = int(rng.choice([7, 3000, 900_000]))
    ids = rng.integers(0, ngr, n).astype(np.uint32)
    vt = VT[int(rng.integers(0, len(VT)))]
    v = orc.random_arr(rng, vt, n, float(rng.choice([0, .05, .5])), int(rng.integers(0, 5)), lo=-1000 if vt not in (orc.UINT8, orc.UINT64) else 0, hi=1000)
    for fn in ("sum", "count", "mean"):
        t0 = time.time()
        a, b = o.hash_aggregate(fn, v, ids, ngr, 0), g.hash_aggregate(fn, v, ids, ngr, 0)
        ok = a.same(b) if a.values.dtype.kind != "f" else (close(a.values, b.values) and a.null_count == b.null_count)
        rep((fn, vt, ngr, n), ok, t0)
    t0 = time.time()
    a, b = o.hash_min_max(v, ids, ngr), g.hash_min_max(v, ids, ngr)
    ok = all(x.same(y) for x, y in zip(a, b))
    rep(("hash_min_max", vt, ngr), ok, t0)
    # unique / value_counts / dictionary_encode
    kt = KT[int(rng.integers(0, len(KT)))]
    a1 = keycol(kt, n, ng, int(rng.integers(0, 5)))
    for name in ("unique", "value_counts", "dictionary_encode"):
        t0 = time.time()
        w, r = getattr(o, name)(a1), getattr(g, name)(a1)
        if isinstance(w, tuple): ok = all((x.same(y) if hasattr(x, "same") else np.array_equal(x, y)) for x, y in zip(w, r))
        else: ok = w.same(r)
        rep((name, kt, ng), ok, t0)
    # scalar aggregate
    t0 = time.time()
    w, r = o.aggregate(v), g.aggregate(v)
    ok = all((abs(w[k] - r[k]) <= 1e-9 * max(1.0, abs(w[k]))) if isinstance(w[k], float) and w[k] == w[k] else (w[k] == r[k] or (w[k] != w[k] and r[k] != r[k])) for k in w)
    rep(("aggregate", vt), ok, t0)
print("bad:", bad)
