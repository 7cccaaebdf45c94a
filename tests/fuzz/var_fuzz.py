import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import numpy as np, orc, gpu_api
import nested as T
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
g = gpu_api.GpuNested()
KINDS = [[T.K_UTF8], [T.K_LARGE_BINARY], [T.K_BINARY], [T.K_UTF8, T.K_PRIM], [T.K_PRIM, T.K_BINARY, T.K_DICT], [T.K_DICT], [T.K_UTF8, T.K_UTF8]]
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    kinds = KINDS[int(rng.integers(0, len(KINDS)))]
    n = int(rng.choice([20_000, 300_000, 700_000]))
    nb = int(rng.integers(1, 4))
    pool = int(rng.choice([30, 5000, 150_000]))
    maxlen = int(rng.choice([3, 40, 200]))
    pn = float(rng.choice([0, .05, .4]))
    t0 = time.time()
    batches = T.rand_key_batches(rng, kinds, n, nb, pn, pool_size=pool, maxlen=maxlen)
    want = T.oracle_group(batches)
    try:
        got = g.group(batches)
        ok = got[1] == want[1] and all(np.array_equal(x, y) for x, y in zip(got[0], want[0])) and all(T.same_where_valid(a, b)[0] for a, b in zip(want[2], got[2]))
    except orc.StatusError as e:
        ok = False; print("status error", e.code, e.msg)
    bad += not ok
    print(it, kinds, n, nb, pool, maxlen, pn, "groups", want[1], "ok" if ok else "MISMATCH", "%.1fs" % (time.time() - t0), flush=True)
print("bad:", bad)
