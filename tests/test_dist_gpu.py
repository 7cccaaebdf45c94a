"""GPU: the multi-rank GroupByMerger code path (side stream, double buffering, dense all-reduce
merge) with TWO ranks sharing one GPU.  RCCL refuses two ranks per device, so the collectives go
through gloo here (CUDA tensors staged by gloo); everything else -- the fused HIP kernel per shard,
the dense scatter, the ordering, the stream/event choreography -- is the code `bench.py --gpus N`
runs under RCCL."""
import datetime
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    """a port nobody listens on right now (bind to 0, read it back) instead of one derived from the pid"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _join_all(procs, seconds):
    """join with a deadline; whatever is still running afterwards is killed so that a stuck rendezvous can never keep
    the test session alive, then the exit codes are checked"""
    try:
        for p in procs:
            p.join(seconds)
    finally:
        stuck = [p for p in procs if p.is_alive()]
        for p in stuck:
            p.kill()
        for p in stuck:
            p.join(10)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as tdist
    tdist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    torch.cuda.set_device(0)
    import orc
    from __graft_entry__ import load_package
    load_package()
    from arrow1_amd import _abi, device, dist as adist
    ctx = device.Context(0)
    rng = np.random.default_rng(0x0ff1ce)
    vals = rng.integers(-100, 101, n)
    keys = rng.integers(0, 50_000, n).astype(np.int32)
    mask = rng.random(n) < 0.5
    per = ((n // world + 63) // 64) * 64
    lo, hi = rank * per, min(n, (rank + 1) * per)

    def up(t, a):
        return device.DeviceArray.from_numpy(t, a)

    merger = adist.GroupByMerger(ctx, rank, world, key_type=_abi.INT32, value_type=_abi.INT64, group_capacity=1 << 17)
    results = []
    for step in range(4):  # several steps: exercises the double buffering and the side stream
        r = merger.filter_groupby(up(_abi.INT64, vals[lo:hi]), up(_abi.INT32, keys[lo:hi]), up(_abi.BOOL, mask[lo:hi]))
        results.append(r)
    merger.finish()
    torch.cuda.synchronize()
    if rank == 0:
        o = orc.Api("orc")
        es, ec, ek, ekv = o.filter_groupby_sum(orc.Arr(orc.INT64, vals), orc.Arr(orc.INT32, keys), orc.Arr(orc.BOOL, mask))
        ok = True
        for r in results[-2:]:  # the two results whose buffers are still live
            k, kv, s, c = r.tensors()
            ok &= r.num_groups == len(es)
            ok &= np.array_equal(s.cpu().numpy(), es) and np.array_equal(c.cpu().numpy(), ec)
            ok &= np.array_equal(k.cpu().numpy(), ek.astype(np.int64)) and bool(kv.all())
        q.put(bool(ok))
    tdist.barrier()
    tdist.destroy_process_group()


def test_two_ranks_one_gpu_dense_merge():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 700_000, q)) for r in range(2)]
    for p in procs:
        p.daemon = True
        p.start()
    _join_all(procs, 300)
    assert q.get(timeout=5) is True


def _replan_worker(rank, world, port, n, q):
    """GroupByMerger over steps whose keys move: step 1's keys on rank 1 jump far outside the dense plan (the step is
    redone on the gather protocol at wait(), the plan is widened), step 2 runs dense again on the widened plan.  Every
    step's merged table equals the single-process GroupBy over both shards."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as tdist
    tdist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    torch.cuda.set_device(0)
    import orc
    from __graft_entry__ import load_package
    load_package()
    from arrow1_amd import _abi, device, dist as adist
    ctx = device.Context(0)
    per = ((n // world + 63) // 64) * 64
    lo, hi = rank * per, min(n, (rank + 1) * per)
    merger = adist.GroupByMerger(ctx, rank, world, key_type=_abi.INT32, value_type=_abi.INT64, group_capacity=1 << 17)
    o = orc.Api("orc")
    ok = True
    for step in range(3):
        rng = np.random.default_rng(50 + step)
        vals = rng.integers(-100, 101, n)
        keys = rng.integers(0, 30_000, n).astype(np.int32)
        if step == 1:
            keys[per:] += 5_000_000
        mask = rng.random(n) < 0.5
        r = merger.filter_groupby(device.DeviceArray.from_numpy(_abi.INT64, vals[lo:hi]), device.DeviceArray.from_numpy(_abi.INT32, keys[lo:hi]),
                                  device.DeviceArray.from_numpy(_abi.BOOL, mask[lo:hi]))
        k, kv, s, c = r.tensors()  # waits: a step that left the plan is redone here, on every rank
        es, ec, ek, ekv = o.filter_groupby_sum(orc.Arr(orc.INT64, vals), orc.Arr(orc.INT32, keys), orc.Arr(orc.BOOL, mask))
        ok &= r.num_groups == len(es) and np.array_equal(s.cpu().numpy(), es) and np.array_equal(c.cpu().numpy(), ec)
        ok &= np.array_equal(k.cpu().numpy().astype(np.int64), ek.astype(np.int64))
    ok &= merger.replans == 1
    merger.finish()
    q.put(bool(ok))
    tdist.barrier()
    tdist.destroy_process_group()


def test_two_ranks_one_gpu_keys_leave_the_dense_plan():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replan_worker, args=(r, 2, port, 400_000, q)) for r in range(2)]
    for p in procs:
        p.daemon = True
        p.start()
    _join_all(procs, 300)
    assert q.get(timeout=5) is True and q.get(timeout=5) is True


def _sort_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as tdist
    tdist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    torch.cuda.set_device(0)
    import orc
    from __graft_entry__ import load_package
    load_package()
    from arrow1_amd import _abi, device, dist as adist
    ctx = device.Context(0)
    rng = np.random.default_rng(0x0ff1ce)
    ok = True
    for kind, tid, desc, null_p in [("float64", _abi.DOUBLE, False, 0.03), ("int64", _abi.INT64, True, 0.0), ("uint64", _abi.UINT64, False, 0.0)]:
        if kind == "float64":
            data = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)
            data[rng.random(n) < 0.01] = np.nan
        elif kind == "uint64":
            data = rng.integers(0, 2 ** 64, n, dtype=np.uint64)
        else:
            data = rng.integers(-5000, 5000, n)
        valid = rng.random(n) >= null_p if null_p else None
        per = ((n // world + 63) // 64) * 64
        lo, hi = rank * per, (min(n, (rank + 1) * per) if rank < world - 1 else n)
        local = data[lo:hi]
        t = torch.from_numpy(local.view(np.int64) if kind == "uint64" else local.copy()).cuda()
        tv = None if valid is None else torch.from_numpy(valid[lo:hi].copy()).cuda()
        out = adist.sharded_sort_indices(adist.SortOps(ctx, tid), t, tv, lo, kind, desc, rank, world).cpu().numpy()
        gathered = [None] * world
        tdist.all_gather_object(gathered, out)
        if rank == 0:
            o = orc.Api("orc")
            want = o.sort_indices([orc.Arr({"float64": orc.DOUBLE, "int64": orc.INT64, "uint64": orc.UINT64}[kind], data, valid)],
                                  orc.DESC if desc else orc.ASC).astype(np.int64)
            ok &= bool(np.array_equal(np.concatenate(gathered), want))
    if rank == 0:
        q.put(bool(ok))
    tdist.barrier()
    tdist.destroy_process_group()


def test_two_ranks_one_gpu_sharded_sort():
    """The sample sort of dist.sharded_sort_indices with the HIP sort as the local operation, two ranks on the
    one device (collectives over gloo): the concatenated result is sort_indices of the whole column."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sort_worker, args=(r, 2, port, 300_000, q)) for r in range(2)]
    for p in procs:
        p.daemon = True
        p.start()
    _join_all(procs, 300)
    assert q.get(timeout=5) is True
