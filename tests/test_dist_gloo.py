"""CPU, world_size 2, gloo: the sharded group-by protocol of arrow-1_amd/dist.py with the
oracle standing in for the kernels.  Checks that row-range shards + all-gather + re-aggregation
reproduce the single-process GroupBy row for row (first-appearance order included)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, null_keys, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    import orc
    from __graft_entry__ import load_package
    load_package()
    from arrow1_amd import dist as adist
    o = orc.Api("orc")
    rng = np.random.default_rng(0x0ff1ce)
    vals = rng.integers(-100, 101, n)
    keys = rng.integers(0, 500, n).astype(np.int32)
    kvalid = rng.random(n) >= (0.05 if null_keys else 0.0)
    mask = rng.random(n) < 0.5
    if rank == 1:
        mask[: n // 2 + 10] |= False
    # contiguous row ranges, multiples of 64 rows
    per = ((n // world + 63) // 64) * 64
    lo, hi = rank * per, min(n, (rank + 1) * per)

    def arrs(sl):
        return (orc.Arr(orc.INT64, vals[sl]), orc.Arr(orc.INT32, keys[sl], kvalid[sl] if null_keys else None),
                orc.Arr(orc.BOOL, mask[sl]))

    s, c, k, kv = o.filter_groupby_sum(*arrs(slice(lo, hi)))
    part = adist.Partial(torch.from_numpy(k), torch.from_numpy(kv), torch.from_numpy(s), torch.from_numpy(c))

    def local_merge(keys_t, kv_t, sums_t, counts_t, live_t):
        karr = orc.Arr(orc.INT32, keys_t.numpy(), kv_t.numpy())
        marr = orc.Arr(orc.BOOL, live_t.numpy())
        s2, _, k2, kv2 = o.filter_groupby_sum(orc.Arr(orc.INT64, sums_t.numpy()), karr, marr)
        c2, _, _, _ = o.filter_groupby_sum(orc.Arr(orc.INT64, counts_t.numpy()), karr, marr)
        return adist.Partial(torch.from_numpy(k2), torch.from_numpy(kv2), torch.from_numpy(s2), torch.from_numpy(c2))

    merged = adist.merge_partials(part, local_merge, world)
    # the dense (all-reduce) protocol on the same partials
    lo_k, hi_k = adist.key_range(part)
    st = adist.DenseState(lo_k - 3, hi_k - lo_k + 1 + 9, torch.int64, torch.device("cpu"))
    dpart, n_present = adist.merge_dense(part, st, rank)
    G = int(n_present.item())
    if rank == 0:
        es, ec, ek, ekv = o.filter_groupby_sum(*arrs(slice(0, n)))
        ok = (np.array_equal(merged.sums.numpy(), es) and np.array_equal(merged.counts.numpy(), ec) and
              np.array_equal(merged.keys.numpy()[ekv], ek[ekv]) and np.array_equal(merged.key_valid.numpy(), ekv))
        ok_dense = (G == len(es) and int(st.bad.item()) == 0 and np.array_equal(dpart.sums[:G].numpy(), es) and
                    np.array_equal(dpart.counts[:G].numpy(), ec) and np.array_equal(dpart.key_valid[:G].numpy(), ekv) and
                    np.array_equal(dpart.keys[:G].numpy()[ekv], ek[ekv].astype(np.int64)))
        q.put(bool(ok and ok_dense))
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("null_keys", [False, True])
def test_sharded_groupby_matches_single_process(null_keys):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + (1 if null_keys else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 20_000, null_keys, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
