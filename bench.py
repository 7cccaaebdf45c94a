#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Arrow compute backend.

Default workload (BASELINE.json metric / SURVEY.md section 8d "headline"): fused
Filter -> GroupBy-sum over 100 M int64 rows, int32 keys uniform over 1 M groups, Bernoulli(0.5)
mask, one step = one pass over the resident batch.  Inputs are generated on the device and are
HBM-resident before the timed region.  Algorithmic bytes = 12.125 B/row (8 value + 4 key + 1/8 mask).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--rows R]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, rows sharded by contiguous row range (weak scaling: every rank
holds --rows rows), per-rank fused kernel, then ONE RCCL exchange for the group-by merge.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from arrow1_amd import _abi, compute, device  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
SEED = 0x0ff1ce         # kRandomSeed, compute/kernels/test_util.h:123

VALUE_RANGE = "small"
VALUE_RANGE_GIVEN = False
NUM_GROUPS = 1_000_000  # --groups
NUM_CHUNKS = 1  # --chunks (sort workloads)
HOT_FRAC = 0.0  # --hot-frac (headline workload: this share of the rows carries ONE key)
KEY_BITS = 32   # --key-bits (headline workload: 64 = int64 key column)
WIDE_KEYS = False  # --wide-keys (with --key-bits 64: the keys span the whole 64-bit range)

WORKLOADS = {
    # name: (description, algorithmic bytes per row as a function of selectivity)
    "filter_groupby_sum_i64": "Filter(mask 50%) -> GroupBy{hash_sum,hash_count}(int64) by int32 key, 1M groups",
    "filter_i64": "Filter int64 + validity (p_null .01), Bernoulli(s) mask, DROP",
    "take_i64": "Take int64 + validity by uint32 indices (monotonic = GetTakeIndices(mask))",
    "take_random_i64": "Take int64 + validity by uint32 indices (uniform random, M = s*N of them)",
    "sort_u64": "sort_indices uint64 full range",
    "sort_f64": "sort_indices float64 U[0,1) + 0.1% NaN + 1% null",
    "groupby_2key_i64": "Filter(mask 50%) -> GroupBy{hash_sum,hash_count}(int64) by TWO int32 key columns (1000 x 1000 groups)",
    "filter_utf8": "Filter of a utf8 column (lengths U[0,16], mean 8 bytes; 32-bit offsets), Bernoulli(s) mask, DROP",
    "sort_2key": "sort_indices of a batch by (int32 key over --groups values ascending, int64 + validity descending)",
    "hash_agg_f64": "GroupBy sum/count over float64 (p_null .01), 1M int32 keys",
    "compare_i64": "greater(int64 array, int64 array)",
    "agg_f64": "count + sum + min_max of float64 + validity (one pass feeds the four scalar aggregates)",
    "mask_and_kleene": "and_kleene(boolean + validity, boolean + validity): the mask-building step of a predicate",
}


def device_bitmap(n, p_true, gen, dev):
    """Bernoulli(p) bitmap of n bits built on the device (LSB-first)."""
    bits = torch.rand(((n + 7) // 8) * 8, device=dev, generator=gen) < p_true
    if n % 8:
        bits[n:] = False
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)
    packed = (bits.view(-1, 8).to(torch.uint8) * w).sum(1, dtype=torch.int32).to(torch.uint8)
    buf = device.alloc_bytes(packed.numel(), dev, zero=True)
    buf[:packed.numel()] = packed
    return buf


def make_inputs(workload, n, dev, rank, sel):
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED + rank)
    d = {}
    if workload in ("filter_groupby_sum_i64",):
        if VALUE_RANGE == "full":  # int64 values over the whole 64-bit range (12-byte compact records instead of 8-byte ones)
            vals = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=gen)
        else:  # int64 uniform in [-100, 100]
            vals = torch.randint(-100, 101, (n,), dtype=torch.int64, device=dev, generator=gen)
        kdt, kid = (torch.int64, _abi.INT64) if KEY_BITS == 64 else (torch.int32, _abi.INT32)
        keys = torch.randint(0, NUM_GROUPS, (n,), dtype=kdt, device=dev, generator=gen)
        if HOT_FRAC > 0:  # a heavy hitter among many distinct keys (skew)
            keys[torch.rand(n, device=dev, generator=gen) < HOT_FRAC] = 7
        if KEY_BITS == 64 and WIDE_KEYS:  # ids / hashes: NUM_GROUPS distinct values spread over the whole 64-bit range
            pool = torch.randint(-2**63, 2**63 - 1, (NUM_GROUPS,), dtype=torch.int64, device=dev, generator=gen)
            keys = pool[keys]
        elif KEY_BITS == 32 and WIDE_KEYS:  # sparse ids: NUM_GROUPS distinct values spread over the int32 range (hashed tables)
            pool = torch.randint(-2**31, 2**31 - 1, (NUM_GROUPS,), dtype=torch.int32, device=dev, generator=gen)
            keys = pool[keys.long()]
        elif KEY_BITS == 64:
            keys += 1_700_000_000_000  # e.g. millisecond timestamps / large ids: wide values, narrow range
        d["values"] = device.DeviceArray(_abi.INT64, vals.view(torch.uint8), None, n)
        d["keys"] = device.DeviceArray(kid, keys.view(torch.uint8), None, n)
        d["mask"] = device.DeviceArray(_abi.BOOL, device_bitmap(n, sel, gen, dev), None, n)
        d["_keep"] = (vals, keys)
    elif workload in ("filter_i64", "take_i64", "take_random_i64"):
        vals = torch.randint(-100, 101, (n,), dtype=torch.int64, device=dev, generator=gen)
        d["values"] = device.DeviceArray(_abi.INT64, vals.view(torch.uint8), device_bitmap(n, 0.99, gen, dev), n, 0, -1)
        d["mask"] = device.DeviceArray(_abi.BOOL, device_bitmap(n, sel, gen, dev), None, n)
        d["_keep"] = (vals,)
    elif workload == "sort_u64":
        vals = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=gen)
        d["values"] = device.DeviceArray(_abi.UINT64, vals.view(torch.uint8), None, n)
        d["_keep"] = (vals,)
    elif workload == "sort_f64":
        vals = torch.rand(n, dtype=torch.float64, device=dev, generator=gen)
        vals[torch.rand(n, device=dev, generator=gen) < 0.001] = float("nan")
        d["values"] = device.DeviceArray(_abi.DOUBLE, vals.view(torch.uint8), device_bitmap(n, 0.99, gen, dev), n, 0, -1)
        d["_keep"] = (vals,)
    elif workload == "groupby_2key_i64":
        vals = torch.randint(-100, 101, (n,), dtype=torch.int64, device=dev, generator=gen)
        k1 = torch.randint(0, 1000, (n,), dtype=torch.int32, device=dev, generator=gen)
        k2 = torch.randint(50_000, 51_000, (n,), dtype=torch.int32, device=dev, generator=gen)
        d["values"] = device.DeviceArray(_abi.INT64, vals.view(torch.uint8), None, n)
        d["k1"] = device.DeviceArray(_abi.INT32, k1.view(torch.uint8), None, n)
        d["k2"] = device.DeviceArray(_abi.INT32, k2.view(torch.uint8), None, n)
        d["mask"] = device.DeviceArray(_abi.BOOL, device_bitmap(n, sel, gen, dev), None, n)
        d["_keep"] = (vals, k1, k2)
    elif workload == "filter_utf8":
        lens = torch.randint(0, 17, (n,), dtype=torch.int32, device=dev, generator=gen)
        off = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        torch.cumsum(lens, 0, dtype=torch.int32, out=off[1:])
        total = int(off[-1].item())
        data = torch.randint(97, 123, (total + 8,), dtype=torch.uint8, device=dev, generator=gen)
        d["values"] = device.DeviceBinaryArray(off.view(torch.uint8), data, None, n)
        d["mask"] = device.DeviceArray(_abi.BOOL, device_bitmap(n, sel, gen, dev), None, n)
        d["_keep"] = (off, data)
        d["_mean_len"] = total / n
    elif workload == "sort_2key":
        if VALUE_RANGE_GIVEN and VALUE_RANGE == "small":  # (--value-range small: the two keys fit one word together, the composite-key sort)
            vals = torch.randint(-100, 101, (n,), dtype=torch.int64, device=dev, generator=gen)
        else:
            vals = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=dev, generator=gen)
        keys = torch.randint(0, NUM_GROUPS, (n,), dtype=torch.int32, device=dev, generator=gen)
        d["values"] = device.DeviceArray(_abi.INT64, vals.view(torch.uint8), device_bitmap(n, 0.99, gen, dev), n, 0, -1)
        d["keys"] = device.DeviceArray(_abi.INT32, keys.view(torch.uint8), None, n)
        d["_keep"] = (vals, keys)
    elif workload == "hash_agg_f64":
        vals = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1e3
        keys = torch.randint(0, NUM_GROUPS, (n,), dtype=torch.int32, device=dev, generator=gen)
        d["values"] = device.DeviceArray(_abi.DOUBLE, vals.view(torch.uint8), device_bitmap(n, 0.99, gen, dev), n, 0, -1)
        d["keys"] = device.DeviceArray(_abi.INT32, keys.view(torch.uint8), None, n)
        d["_keep"] = (vals, keys)
    elif workload == "agg_f64":
        vals = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1e3
        d["values"] = device.DeviceArray(_abi.DOUBLE, vals.view(torch.uint8), device_bitmap(n, 0.99, gen, dev), n, 0, -1)
        d["_keep"] = (vals,)
    elif workload == "mask_and_kleene":
        d["a"] = device.DeviceArray(_abi.BOOL, device_bitmap(n, 0.5, gen, dev), device_bitmap(n, 0.95, gen, dev), n, 0, -1)
        d["b"] = device.DeviceArray(_abi.BOOL, device_bitmap(n, 0.5, gen, dev), device_bitmap(n, 0.95, gen, dev), n, 0, -1)
    elif workload == "compare_i64":
        a = torch.randint(-100, 101, (n,), dtype=torch.int64, device=dev, generator=gen)
        b = torch.randint(-100, 101, (n,), dtype=torch.int64, device=dev, generator=gen)
        d["a"] = device.DeviceArray(_abi.INT64, a.view(torch.uint8), None, n)
        d["b"] = device.DeviceArray(_abi.INT64, b.view(torch.uint8), None, n)
        d["_keep"] = (a, b)
    else:
        raise SystemExit("unknown workload " + workload)
    return d


def bytes_per_row(workload, sel):
    return {"filter_groupby_sum_i64": 12.125, "filter_i64": 8.25 + 8.125 * sel, "take_i64": 20.25 * sel,
            "take_random_i64": 20.25 * sel, "sort_u64": 16.0, "sort_f64": 16.0, "hash_agg_f64": 12.125, "compare_i64": 16.25,
            "sort_2key": 20.125,  # both key columns + validity in, uint64 indices out
            "groupby_2key_i64": 16.125,  # values + two key columns + mask
            "filter_utf8": 4 + 8 + 0.125 + sel * (4 + 8),  # offsets + bytes (mean 8) + mask in, offsets + bytes out
            "mask_and_kleene": 0.75, "agg_f64": 8.125}[workload]  # 2 x (data + validity) bits in, data + validity bits out


def group_capacity():
    """Room for --groups groups (2 M at the default)."""
    return max(1 << 21, 1 << (NUM_GROUPS + 1).bit_length())


def make_step(workload, ctx, inp, sel, world, rank):
    """Returns a zero-argument callable running ONE step on this rank's shard."""
    if workload == "filter_groupby_sum_i64":
        bufs = compute.GroupByBuffers(ctx.device, group_capacity(), _abi.INT64 if KEY_BITS == 64 else _abi.INT32)
        if world == 1:
            return lambda: compute.filter_groupby(ctx, inp["values"], inp["keys"], inp["mask"], buffers=bufs)
        from arrow1_amd import dist as adist
        merger = adist.GroupByMerger(ctx, rank, world, key_type=_abi.INT32, value_type=_abi.INT64, group_capacity=1 << 21)
        step = lambda: merger.filter_groupby(inp["values"], inp["keys"], inp["mask"])  # noqa: E731
        step.finish = merger.finish  # merges of the last steps still in flight on the side stream
        return step
    if workload == "hash_agg_f64":
        bufs = compute.GroupByBuffers(ctx.device, group_capacity(), _abi.INT32)
        return lambda: compute.filter_groupby(ctx, inp["values"], inp["keys"], None, buffers=bufs)
    if workload == "filter_i64":
        return lambda: compute.filter(ctx, inp["values"], inp["mask"])
    if workload == "take_i64":
        idx = compute.take_indices(ctx, inp["mask"])
        return lambda: compute.take(ctx, inp["values"], idx)
    if workload == "take_random_i64":
        m = int(inp["values"].length * sel)
        gen = torch.Generator(device=inp["values"].values.device)
        gen.manual_seed(SEED + 77 + rank)
        ridx = torch.randint(0, inp["values"].length, (m,), dtype=torch.int64, device=inp["values"].values.device,
                             generator=gen).to(torch.int32)  # values < 2^31: same bits as uint32
        idx = device.DeviceArray(_abi.UINT32, ridx.view(torch.uint8), None, m)
        inp["_keep_idx"] = ridx
        return lambda: compute.take(ctx, inp["values"], idx)
    if workload in ("sort_u64", "sort_f64") and world > 1:
        # one global sort of world x rows: the sample sort of dist.sharded_sort_indices (one all-to-all)
        from arrow1_amd import dist as adist
        vals = inp["_keep"][0]
        valid = None
        if inp["values"].validity is not None:
            valid = adist._unpack_bits(inp["values"].validity, inp["values"].length)
        ops = adist.SortOps(ctx, inp["values"].type_id)
        kind = "uint64" if workload == "sort_u64" else "float64"
        return lambda: adist.sharded_sort_indices(ops, vals, valid, rank * inp["values"].length, kind, False, rank, world)
    if workload in ("sort_u64", "sort_f64"):
        chunks = inp["values"]
        if NUM_CHUNKS > 1:  # the same rows as NUM_CHUNKS equal slices: a ChunkedArray (vector_sort.cc:657-832)
            per = (chunks.length + NUM_CHUNKS - 1) // NUM_CHUNKS
            chunks = [chunks.slice(i * per, min(per, chunks.length - i * per)) for i in range(NUM_CHUNKS)]
        return lambda: compute.sort_indices(ctx, chunks)
    if workload == "groupby_2key_i64":
        return lambda: compute.filter_groupby_multi(ctx, inp["values"], [inp["k1"], inp["k2"]], inp["mask"], 1 << 21)
    if workload == "filter_utf8":
        return lambda: compute.binary_filter(ctx, inp["values"], inp["mask"])
    if workload == "sort_2key":
        return lambda: compute.sort_indices_multi(ctx, [inp["keys"], inp["values"]], [_abi.ASCENDING, _abi.DESCENDING])
    if workload == "compare_i64":
        return lambda: compute.compare(ctx, _abi.GT, inp["a"], inp["b"])
    if workload == "agg_f64":
        return lambda: compute.aggregate(ctx, inp["values"])
    if workload == "mask_and_kleene":
        return lambda: compute.boolean(ctx, _abi.AND_KLEENE, inp["a"], inp["b"])
    raise SystemExit("unknown workload " + workload)


def cpu_baseline(workload, sel):
    """Time the CPU path on a BOUNDED sample of the same workload on this box's host cores:
    the real reference (oracle/_ref, kind 'reference') when it travelled with the repo, else
    the oracle restatement (kind 'port').  1 thread: the reference compute layer is
    single-threaded (compute/exec.h:86-92)."""
    import orc
    kind = "reference" if orc.have_ref() else "port"
    api = orc.Api("ref" if kind == "reference" else "orc")
    rng = np.random.default_rng(SEED)
    n = {"sort_u64": 2_000_000, "sort_f64": 2_000_000, "sort_2key": 2_000_000}.get(workload, 8_000_000)

    def nullable(type_id, data, p_null):
        return orc.Arr(type_id, data, rng.random(len(data)) >= p_null)

    if workload == "filter_groupby_sum_i64":
        vals = orc.Arr(orc.INT64, rng.integers(-100, 101, n))
        keys = orc.Arr(orc.INT32, rng.integers(0, NUM_GROUPS, n))
        mask = orc.Arr(orc.BOOL, rng.random(n) < sel)
        call = lambda: api.filter_groupby_sum(vals, keys, mask)  # noqa: E731
        what = "Filter+Filter+GroupBy{hash_sum,hash_count}, 1M keys, mask %.0f%%" % (sel * 100)
    elif workload == "hash_agg_f64":
        vals = nullable(orc.DOUBLE, rng.random(n) * 1e3, 0.01)
        keys = orc.Arr(orc.INT32, rng.integers(0, NUM_GROUPS, n))
        mask = orc.Arr(orc.BOOL, np.ones(n, bool))
        call = lambda: api.filter_groupby_sum(vals, keys, mask)  # noqa: E731
        what = "GroupBy{hash_sum,hash_count} float64, 1M keys (through an all-true Filter)"
    elif workload == "filter_i64":
        vals = nullable(orc.INT64, rng.integers(-100, 101, n), 0.01)
        mask = orc.Arr(orc.BOOL, rng.random(n) < sel)
        call = lambda: api.filter(vals, mask)  # noqa: E731
        what = "filter int64+validity, mask %.0f%%" % (sel * 100)
    elif workload in ("take_i64", "take_random_i64"):
        vals = nullable(orc.INT64, rng.integers(-100, 101, n), 0.01)
        if workload == "take_i64":
            idx = api.take_indices(orc.Arr(orc.BOOL, rng.random(n) < sel))
            idx = orc.Arr(orc.UINT32, idx.values)
        else:
            idx = orc.Arr(orc.UINT32, rng.integers(0, n, int(n * sel)))
        call = lambda: api.take(vals, idx)  # noqa: E731
        what = "take int64+validity by %d uint32 indices" % idx.length
    elif workload == "sort_u64":
        vals = orc.Arr(orc.UINT64, rng.integers(0, 2**64, n, dtype=np.uint64))
        call = lambda: api.sort_indices([vals])  # noqa: E731
        what = "sort_indices uint64"
    elif workload == "sort_f64":
        data = rng.random(n)
        data[rng.random(n) < 0.001] = np.nan
        vals = nullable(orc.DOUBLE, data, 0.01)
        call = lambda: api.sort_indices([vals])  # noqa: E731
        what = "sort_indices float64 (0.1% NaN, 1% null)"
    elif workload == "filter_utf8":
        n = 4_000_000
        lens = rng.integers(0, 17, n)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        v = orc.BinArr.__new__(orc.BinArr)
        v.items, v.offsets, v.data = None, off, rng.integers(97, 123, int(off[-1]) + 8).astype(np.uint8)
        v.valid, v.vbits, v.offset, v.length, v.null_count = None, None, 0, n, 0
        mask = orc.Arr(orc.BOOL, rng.random(n) < sel)
        call = lambda: api.binary_filter(v, mask)  # noqa: E731
        what = "filter utf8 (mean 8 bytes), mask %.0f%%" % (sel * 100)
    elif workload == "sort_2key":
        vals = nullable(orc.INT64, rng.integers(-2**62, 2**62, n), 0.01)
        keys = orc.Arr(orc.INT32, rng.integers(0, NUM_GROUPS, n))
        call = lambda: api.sort_indices_multi([keys, vals], [orc.ASC, orc.DESC])  # noqa: E731
        what = "sort_indices(RecordBatch, {int32 asc, int64 desc}) -- RadixRecordBatchSorter"
    elif workload == "agg_f64":
        vals = nullable(orc.DOUBLE, rng.random(n) * 1e3, 0.01)
        call = lambda: api.aggregate(vals)  # noqa: E731
        what = "count + sum + min_max of float64 (three CallFunctions on the reference)"
    elif workload == "mask_and_kleene":
        a = orc.Arr(orc.BOOL, rng.random(n) < .5, rng.random(n) < .95)
        b = orc.Arr(orc.BOOL, rng.random(n) < .5, rng.random(n) < .95)
        call = lambda: api.boolean(orc.AND_KLEENE, a, b)  # noqa: E731
        what = "and_kleene(boolean, boolean), 5% nulls each"
    elif workload == "compare_i64":
        a, b = orc.Arr(orc.INT64, rng.integers(-100, 101, n)), orc.Arr(orc.INT64, rng.integers(-100, 101, n))
        call = lambda: api.compare(orc.GT, a, b)  # noqa: E731
        what = "greater(int64, int64)"
    else:
        return None
    t0 = time.time()
    reps = 0
    while True:
        call()
        reps += 1
        if time.time() - t0 > 10.0 or reps >= 20:
            break
    dt = (time.time() - t0) / reps
    out = {"value": n / dt, "unit": "rows/s", "cores": 1, "kind": kind,
           "sample": "%d rows (same generator family), %d reps of %s" % (n, reps, what)}
    # the same function from `cores` host threads, each on a contiguous row range of the sample (the
    # reference's compute layer is single-threaded; this is what a caller gets by sharding rows over the host's cores).
    # Group-by partials are merged by key on the host inside the timed region.  ctypes releases the GIL for the calls.
    if workload in ("filter_groupby_sum_i64", "hash_agg_f64", "filter_i64", "compare_i64"):
        import concurrent.futures as cf
        cores = max(1, min(os.cpu_count() or 1, 16))  # the box's CPU share for one GPU
        per = (n + cores - 1) // cores
        cuts = [(i * per, min(n, (i + 1) * per)) for i in range(cores) if i * per < n]

        def sl(a, lo, hi):
            return orc.Arr(a.type_id, a.data[lo:hi], None if a.valid is None else a.valid[lo:hi])
        if workload in ("filter_groupby_sum_i64", "hash_agg_f64"):
            parts = [(sl(vals, lo, hi), sl(keys, lo, hi), sl(mask, lo, hi)) for lo, hi in cuts]
            apis = [orc.Api("ref" if kind == "reference" else "orc") for _ in cuts]

            def run(i):
                return apis[i].filter_groupby_sum(*parts[i])

            def merge(rs):
                allk = np.concatenate([r[2] for r in rs])
                uk, inv = np.unique(allk, return_inverse=True)
                sums = np.zeros(len(uk), rs[0][0].dtype)
                cnts = np.zeros(len(uk), np.int64)
                np.add.at(sums, inv, np.concatenate([r[0] for r in rs]))
                np.add.at(cnts, inv, np.concatenate([r[1] for r in rs]))
                return sums, cnts
        else:
            if workload == "filter_i64":
                parts = [(sl(vals, lo, hi), sl(mask, lo, hi)) for lo, hi in cuts]
                fn = "filter"
            else:
                parts = [(orc.GT, sl(a, lo, hi), sl(b, lo, hi)) for lo, hi in cuts]
                fn = "compare"
            apis = [orc.Api("ref" if kind == "reference" else "orc") for _ in cuts]

            def run(i):
                return getattr(apis[i], fn)(*parts[i])

            def merge(rs):
                return rs
        with cf.ThreadPoolExecutor(len(cuts)) as ex:
            list(ex.map(run, range(len(cuts))))  # warm
            t0 = time.time()
            reps = 0
            while True:
                merge(list(ex.map(run, range(len(cuts)))))
                reps += 1
                if time.time() - t0 > 6.0 or reps >= 20:
                    break
            dts = (time.time() - t0) / reps
        out["row_range_sharded"] = {"value": n / dts, "unit": "rows/s", "cores": len(cuts), "host_cpus": os.cpu_count(),
                                    "note": "%d threads x contiguous row ranges of the same sample%s" %
                                            (len(cuts), ", partials merged by key on the host" if "groupby" in workload or "agg" in workload else "")}
    return out


def _timed(fn, reps):
    """median of `reps` HIP-event timings of fn() on the current stream, in ms"""
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def headline_extras(ctx, inp, n, dev, sel):
    """The same workload off its steady state, measured in this process OUTSIDE the timed region (the headline `value` is
    the steady state of repeated calls of one shape, whose records the context's hint makes compact):
      cold_ms_per_step         a call that takes nothing over from its predecessor (AMDARROW_GROUPBY_NO_HINT=1: what the
                               first call of a shape, or a caller whose data change shape every call, sees)
      wide_values_ms_per_step  int64 values over the whole 64-bit range (12-byte records instead of 8-byte ones), steady state"""
    out = {}
    bufs = compute.GroupByBuffers(ctx.device, group_capacity(), _abi.INT32)
    call = lambda: compute.filter_groupby(ctx, inp["values"], inp["keys"], inp["mask"], buffers=bufs)  # noqa: E731
    os.environ["AMDARROW_GROUPBY_NO_HINT"] = "1"
    try:
        call()
        out["cold_ms_per_step"] = _timed(call, 7)
    finally:
        del os.environ["AMDARROW_GROUPBY_NO_HINT"]
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED + 1000)
    wide = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=gen)
    warr = device.DeviceArray(_abi.INT64, wide.view(torch.uint8), None, n)
    wcall = lambda: compute.filter_groupby(ctx, warr, inp["keys"], inp["mask"], buffers=bufs)  # noqa: E731
    for _ in range(3):
        wcall()
    out["wide_values_ms_per_step"] = _timed(wcall, 7)
    call()  # (leave the context's hint as the timed loop left it)
    return out


def transfer_line(workload, inp, n, dev, res, step_s):
    """Measured H2D / D2H next to the kernel time (reported separately, never part of `value`): the step's
    input buffers from pinned host memory, and its result back."""
    in_bytes = 0
    for k, v in inp.items():
        if k.startswith("_"):
            continue
        for t in (getattr(v, "values", None), getattr(v, "validity", None), getattr(v, "offsets", None), getattr(v, "data", None)):
            if torch.is_tensor(t):
                in_bytes += t.numel() * t.element_size()
    in_bytes = min(in_bytes, 2 << 30)
    host = torch.empty(in_bytes, dtype=torch.uint8).pin_memory()
    devbuf = torch.empty(in_bytes, dtype=torch.uint8, device=dev)
    h2d = _timed(lambda: devbuf.copy_(host, non_blocking=True), 3)
    out_bytes = 0
    if hasattr(res, "num_groups") and hasattr(res, "sums"):
        out_bytes = int(res.num_groups) * 20
    elif hasattr(res, "length"):
        out_bytes = min(int(res.length) * 8, in_bytes)
    d2h = _timed(lambda: host[:max(out_bytes, 1)].copy_(devbuf[:max(out_bytes, 1)], non_blocking=True), 3) if out_bytes else 0.0
    return {"h2d_ms": h2d, "h2d_bytes": in_bytes, "h2d_gbps": in_bytes / h2d / 1e6, "d2h_ms": d2h, "d2h_bytes": out_bytes,
            "rows_per_s_with_transfers": n / (step_s + (h2d + d2h) / 1e3),
            "note": "pinned host memory over PCIe, not overlapped with the kernels; `value` counts HBM-resident inputs only"}


def pmc_traffic(workload):
    """HBM bytes per step from the committed rocprofv3 --pmc summary of this very command
    (profiles/run_pmc.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 correction applied
    there): sum over the kernels of one step.  None when no summary is committed for the workload."""
    if workload != "filter_groupby_sum_i64":
        return None, None
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_headline_hbm_pmc.txt")))
    if not found:
        return None, None
    path = found[-1]  # the latest round's collection
    total_kb = 0.0
    for line in open(path):
        parts = line.split()
        if parts[:1] == ["PER_STEP_STEADY"]:  # FETCH_SIZE and WRITE_SIZE, the kernels of a steady-state step
            total_kb += float(parts[2])
    if total_kb == 0.0:
        return None, None
    return total_kb * 1024.0, os.path.relpath(path, ROOT)


def dominant_kernel(workload, algorithmic_bytes):
    """The op's longest kernel in the committed `rocprofv3 --kernel-trace --stats` summary of this command
    (profiles/rNN_<workload>_kernel_stats.csv): name, average duration, and the algorithmic bytes of a step over that
    duration alone (the op's other kernels not counted -- `frac` above counts the whole step)."""
    import csv
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_kernel_stats.csv" % workload)))
    if not found:
        return None
    best = None
    for r in csv.DictReader(open(found[-1])):
        if "amdarrow::" in r["Name"] and int(r["Calls"]) >= 5 and (best is None or float(r["AverageNs"]) > float(best["AverageNs"])):
            best = r
    if best is None:
        return None
    us = float(best["AverageNs"]) / 1e3
    gbps = algorithmic_bytes / (us * 1e-6) / 1e9
    return {"name": best["Name"].split("(")[0].replace("void ", ""), "avg_us": us, "achieved": gbps, "frac": gbps / HBM_PEAK_GBPS,
            "source": os.path.relpath(found[-1], ROOT)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="filter_groupby_sum_i64", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU")
    ap.add_argument("--selectivity", type=float, default=0.5)
    ap.add_argument("--groups", type=int, default=1_000_000, help="distinct group keys (tuning aid; the headline is 1M)")
    ap.add_argument("--hot-frac", type=float, default=0.0, help="headline workload: share of rows that carry one hot key")
    ap.add_argument("--key-bits", type=int, default=32, choices=[32, 64], help="headline workload: width of the key column")
    ap.add_argument("--wide-keys", action="store_true", help="distinct keys spread over the whole range of the key type (64-bit: the wide-key path; 32-bit: hashed partition tables)")
    ap.add_argument("--chunks", type=int, default=1, help="sort workloads: present the rows as this many equal chunks")
    ap.add_argument("--value-range", default=None, choices=["small", "full"],
                    help="headline workload: int64 values uniform in [-100, 100] (SURVEY 8d; the default) or over the whole 64-bit range; "
                         "sort_2key: 'small' makes the two keys fit one 64-bit word together (default: values over 63 bits)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cold / wide-value legs and the H2D / D2H line (profiling runs)")
    args = ap.parse_args()
    global NUM_GROUPS, NUM_CHUNKS, KEY_BITS, HOT_FRAC, WIDE_KEYS, VALUE_RANGE, VALUE_RANGE_GIVEN
    VALUE_RANGE_GIVEN = args.value_range is not None
    VALUE_RANGE = args.value_range or "small"
    WIDE_KEYS = args.wide_keys
    HOT_FRAC = args.hot_frac
    KEY_BITS = args.key_bits
    NUM_GROUPS = args.groups
    NUM_CHUNKS = args.chunks

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tdist.init_process_group("nccl", device_id=dev)

    ctx = device.Context(local_rank)
    n = args.rows
    sel = args.selectivity
    inp = make_inputs(args.workload, n, dev, rank, sel)
    step = make_step(args.workload, ctx, inp, sel, world, rank)

    finish = getattr(step, "finish", lambda: None)
    for _ in range(args.warmup):
        step()
    finish()
    torch.cuda.synchronize()
    if world > 1:
        tdist.barrier()
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]  # one per step: median / min
    t0 = time.perf_counter()
    ev0.record()  # the kernels run on torch's current stream (device.Context binds it)
    for i in range(args.steps):
        res = step()
        marks[i].record()
    finish()  # every step's merged result is complete inside the timed region
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        tdist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    per_step = sorted((marks[i - 1] if i else ev0).elapsed_time(marks[i]) for i in range(args.steps))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_rows = n * world * args.steps
        value = total_rows / elapsed
        bpr = bytes_per_row(args.workload, sel)
        step_s = dev_ms / 1e3 / args.steps  # HIP-event time of one step's launches on the kernel stream
        achieved = bpr * n / step_s / 1e9
        out = {
            "metric": "rows/sec + achieved HBM GB/s, Filter+GroupBy-sum 100M int64 @1/2/4/8 GPU",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_median": per_step[len(per_step) // 2], "ms_per_step_min": per_step[0],
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"filter_groupby_sum_i64": "int64", "filter_i64": "u64", "take_i64": "u64",
                                            "take_random_i64": "u64", "sort_u64": "u64", "sort_f64": "f64", "hash_agg_f64": "f64",
                                            "compare_i64": "int64", "mask_and_kleene": "u64", "agg_f64": "f64", "sort_2key": "u64", "filter_utf8": "u8", "groupby_2key_i64": "int64"}[args.workload],
            "data": "synthetic (device-generated, seed 0x0ff1ce + rank; HBM-resident before timing)",
            "config": {"workload": args.workload + ": " + WORKLOADS[args.workload], "rows_per_gpu": n,
                       "selectivity": sel, "groups": NUM_GROUPS, "sharding": "row-range, %d shard(s)" % world,
                       # weak scaling: every rank holds --rows rows whatever N is (BASELINE config 5's 1 B rows on 8 GPUs is
                       # `--gpus 8 --rows 125000000`; the default keeps the per-GPU work of the N = 1 line)
                       "total_rows": n * world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "note": "algorithmic %.3f B/row x %d rows / HIP-event time per step (%.3f ms) on the kernel stream; "
                                 "whole step (all kernels of the op), rank 0" % (bpr, n, step_s * 1e3)},
        }
        headline_default = (args.workload == "filter_groupby_sum_i64" and n == 100_000_000 and NUM_GROUPS == 1_000_000 and sel == 0.5
                            and VALUE_RANGE == "small" and KEY_BITS == 32 and not WIDE_KEYS and HOT_FRAC == 0.0)
        traffic, src = pmc_traffic(args.workload)
        if traffic is not None and headline_default:
            out["roofline"]["traffic"] = traffic
            out["roofline"]["traffic_from"] = "committed profile (%s): the builder's box, not this run" % src
            out["roofline"]["note"] += "; traffic = FETCH_SIZE + WRITE_SIZE bytes per steady-state step over all kernels of the op, from %s" % src
        if headline_default:
            dom = dominant_kernel(args.workload, bpr * n)
            if dom:
                dom["from"] = "committed profile: the builder's box, not this run"
                out["roofline"]["dominant_kernel"] = dom
        if hasattr(res, "num_groups"):
            out["config"]["result_groups"] = int(res.num_groups)
        if args.workload in ("filter_groupby_sum_i64", "hash_agg_f64", "groupby_2key_i64"):
            import ctypes as C
            gp, gw = C.c_int(-1), C.c_int(-1)
            ctx.lib.amdarrow_groupby_last_path(ctx.handle, C.byref(gp), C.byref(gw))
            out["config"]["groupby_path"] = {3: "per-workgroup LDS pre-aggregation", 2: "one-pass partition", 1: "two-pass partition",
                                             0: "global table"}.get(gp.value, "?") + (", %d-byte records" % (4 * gw.value) if gw.value else "")
        if headline_default and world == 1 and not args.no_extras:
            out["config"].update(headline_extras(ctx, inp, n, dev, sel))
        if world == 1 and not args.no_extras:
            out["transfer"] = transfer_line(args.workload, inp, n, dev, res, elapsed / args.steps)
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, sel)
            except Exception as e:  # the baseline is a reported extra, never the measured value
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out))
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
