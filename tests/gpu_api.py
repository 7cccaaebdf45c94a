"""The HIP path behind the same numpy-in / numpy-out interface as orc.Api, so parity tests
read `gpu.filter(v, f) == oracle.filter(v, f)`.  Every call goes through the C ABI."""
import numpy as np

import orc
from __graft_entry__ import load_package

load_package()
from arrow1_amd import _abi, compute, device  # noqa: E402


class _Out:
    def __init__(self, length, null_count, has_validity):
        self.length, self.null_count, self.has_validity = length, null_count, has_validity


def up(a):
    return device.DeviceArray.from_numpy(a.type_id, a.data, a.valid, a.offset, a.length, a.null_count)


def down(d, always_validity_bits=False):
    """DeviceArray -> orc.Result"""
    n = d.length
    vals = d.values_numpy()
    if d.validity is not None:
        bits = d.validity[:(n + 7) // 8].cpu().numpy().copy()
    else:
        bits = np.zeros((n + 7) // 8, np.uint8)
    return orc.Result(vals, bits, _Out(n, d.null_count, d.validity is not None), d.type_id)


class GpuApi:
    prefix = "gpu"

    def __init__(self):
        self.ctx = device.Context(0)

    def _wrap(self, fn):
        try:
            return fn()
        except _abi.AmdArrowError as e:
            raise orc.StatusError(e.code, e.msg)

    def filter_output_size(self, filt, null_sel=orc.DROP):
        return self._wrap(lambda: compute.filter_output_size(self.ctx, up(filt), null_sel))

    def filter(self, values, filt, null_sel=orc.DROP):
        return self._wrap(lambda: down(compute.filter(self.ctx, up(values), up(filt), null_sel)))

    def take_indices(self, filt, null_sel=orc.DROP):
        return self._wrap(lambda: down(compute.take_indices(self.ctx, up(filt), null_sel)))

    def take(self, values, indices, boundscheck=True):
        return self._wrap(lambda: down(compute.take(self.ctx, up(values), up(indices), boundscheck)))

    def sort_indices(self, chunks, order=orc.ASC):
        if isinstance(chunks, orc.Arr):
            chunks = [chunks]
        return self._wrap(lambda: compute.sort_indices(self.ctx, [up(c) for c in chunks], order).values_numpy())

    # ---- variable-width Take / Filter ----
    @staticmethod
    def _up_bin(b):
        return device.DeviceBinaryArray.from_items(b.items, b.offset, b.length, b.null_count if b.vbits is not None else 0)

    @staticmethod
    def _down_bin(d):
        n = d.length
        off = d.offsets[:(n + 1) * 4].cpu().numpy().view(np.int32).copy()
        data = d.data[:int(off[-1]) if n else 0].cpu().numpy().copy()
        bits = d.validity[:(n + 7) // 8].cpu().numpy().copy() if d.validity is not None else np.packbits(np.ones(n, bool), bitorder="little")
        return orc.BinResult(off, data, np.concatenate([bits, np.zeros(8, np.uint8)]), _Out(n, d.null_count, d.validity is not None))

    def binary_concatenate(self, chunks):
        return self._wrap(lambda: self._down_bin(compute.binary_concatenate(self.ctx, [self._up_bin(c) for c in chunks])))

    def binary_take(self, values, indices, boundscheck=True):
        return self._wrap(lambda: self._down_bin(compute.binary_take(self.ctx, self._up_bin(values), up(indices), boundscheck)))

    def binary_filter(self, values, filt, null_sel=orc.DROP):
        return self._wrap(lambda: self._down_bin(compute.binary_filter(self.ctx, self._up_bin(values), up(filt), null_sel)))

    # ---- SURVEY 8f.4 ----
    def unique(self, a):
        return self._wrap(lambda: down(compute.unique(self.ctx, up(a))))

    def value_counts(self, a):
        def run():
            u, c = compute.value_counts(self.ctx, up(a))
            return down(u), c.values_numpy()
        return self._wrap(run)

    def dictionary_encode(self, a, null_encoding=1):
        def run():
            i, d = compute.dictionary_encode(self.ctx, up(a), null_encoding)
            r = down(i)
            if i.validity is None:
                r.validity_bits = np.packbits(np.ones(i.length, bool), bitorder="little")
            return r, down(d)
        return self._wrap(run)

    def hash_min_max(self, values, ids, num_groups, null_handling=0, split=None):
        """split: consume the batch as two batches cut at that row (state carried across batches)"""
        import torch

        def run():
            st = compute.HashMinMax(self.ctx, values.type_id, "cuda", null_handling)
            dids = torch.from_numpy(np.ascontiguousarray(ids, np.uint32).view(np.int32)).cuda()
            if num_groups == 0 and values.length == 0:
                st._reserve(0)
            cuts = [0, values.length] if not split else [0, split, values.length]
            for a, b in zip(cuts[:-1], cuts[1:]):
                part = orc.Arr(values.type_id, values.data, values.valid, values.offset + a, b - a,
                               values.null_count if values.null_count in (0, -1) else -1)
                # groups known so far: in a real GroupBy the grouper reports them batch by batch
                known = num_groups if b == values.length else (int(ids[:b].max()) + 1 if b > 0 else 0)
                st.consume(up(part), dids[a:b].contiguous(), known)
            if st.num_groups < num_groups:
                st._reserve(num_groups)
            mn, mx = st.finalize()
            return down(mn), down(mx)
        return self._wrap(run)

    def sort_indices_multi(self, keys, orders):
        keys = [[k] if isinstance(k, orc.Arr) else k for k in keys]
        return self._wrap(lambda: compute.sort_indices_multi(self.ctx, [[up(c) for c in k] for k in keys],
                                                             orders).values_numpy())

    def concatenate(self, chunks):
        return self._wrap(lambda: down(compute.concatenate(self.ctx, [up(c) for c in chunks])))

    def compare(self, op, lhs, rhs):
        f = lambda x: up(x) if isinstance(x, orc.Arr) else x  # noqa: E731
        return self._wrap(lambda: down(compute.compare(self.ctx, op, f(lhs), f(rhs))))

    def arith(self, op, lhs, rhs=None):
        f = lambda x: up(x) if isinstance(x, orc.Arr) else x  # noqa: E731
        return self._wrap(lambda: down(compute.arith(self.ctx, op, f(lhs), f(rhs))))

    def make_groupings(self, ids, num_groups):
        def run():
            off, idx = compute.make_groupings(self.ctx, up(orc.Arr(orc.UINT32, np.asarray(ids, np.uint32))), num_groups)
            return off.cpu().numpy(), idx.cpu().numpy()
        return self._wrap(run)

    def common_numeric(self, type_a, type_b):
        t = compute.common_numeric(type_a, type_b)
        return -1 if t is None else t

    def cast(self, a, to_type, safe=True):
        return self._wrap(lambda: compute.cast(self.ctx, up(a), to_type, safe).values_numpy()[:a.length])

    # ---- mask-building scalars ----
    @staticmethod
    def _down_full(d):
        """like down(), with the validity bits the library wrote even when has_validity == 0"""
        r = down(d)
        vb = getattr(d, "validity_buffer", None)
        if vb is not None:
            r.validity_bits = vb[:(d.length + 7) // 8].cpu().numpy().copy()
        return r

    def boolean(self, op, lhs, rhs=None):
        f = lambda x: up(x) if isinstance(x, orc.Arr) else x  # noqa: E731
        return self._wrap(lambda: self._down_full(compute.boolean(self.ctx, op, f(lhs), f(rhs))))

    def is_valid(self, a):
        return self._wrap(lambda: down(compute.is_valid(self.ctx, up(a))))

    def is_null(self, a):
        return self._wrap(lambda: down(compute.is_null(self.ctx, up(a))))

    def is_nan(self, a):
        return self._wrap(lambda: self._down_full(compute.is_nan(self.ctx, up(a))))

    def fill_null(self, a, fill):
        return self._wrap(lambda: self._down_full(compute.fill_null(self.ctx, up(a), fill)))

    def aggregate(self, a):
        r = self._wrap(lambda: compute.aggregate(self.ctx, up(a)))
        return orc.agg_to_dict(a.type_id, *r)

    def group_ids(self, key_batches):
        g = compute.Grouper(self.ctx, [k.type_id for k in key_batches[0]])
        try:
            ids = [self._wrap(lambda b=b: g.consume([up(k) for k in b]).values_numpy()) for b in key_batches]
            ng = g.num_groups()
            uniq = []
            for k in range(len(key_batches[0])):
                u = g.uniques(k)
                uniq.append((u.values_numpy(), u.validity_numpy(), u.null_count))
            return ids, ng, uniq
        finally:
            g.close()

    def hash_aggregate(self, func, values, ids, num_groups, count_mode=0):
        import torch
        fid = {"sum": 0, "count": 1, "mean": 2}[func]
        t = torch.from_numpy(np.ascontiguousarray(ids, np.uint32).view(np.int32)).cuda()
        pad = torch.zeros(16, dtype=torch.int32, device="cuda")
        t = torch.cat([t, pad])
        d = self._wrap(lambda: compute.hash_aggregate(self.ctx, fid, up(values), t, num_groups, count_mode))
        r = down(d)
        if d.validity is None:  # same convention as orc.Api: bitmap shown as all ones when absent
            bits = np.packbits(np.ones(num_groups, bool), bitorder="little")
            r.validity_bits = bits
        return r

    def filter_groupby_sum(self, values, keys, mask):
        r = self._wrap(lambda: compute.filter_groupby(self.ctx, up(values), up(keys), up(mask) if mask is not None else None))
        return r.sums_numpy(), r.counts_numpy(), r.keys_numpy(), r.key_validity_numpy()


# ---- round 3: the non-primitive array_take / array_filter kernels (tests/nested.py trees <-> arrow1_amd.nested) ----
def up_nested(a):
    import torch
    import nested as T
    from arrow1_amd import nested as P

    def buf(x):
        if x is None:
            return None
        raw = np.packbits(x, bitorder="little") if x.dtype == bool else np.ascontiguousarray(x).view(np.uint8).ravel()
        t = device.alloc_bytes(len(raw), "cuda", zero=True)
        t[:len(raw)] = torch.from_numpy(raw.copy())
        return t
    return P.DeviceNested(a.kind, a.length, a.offset, buf(a.valid), a.null_count, buf(a.buf1), buf(a.buf2),
                          [up_nested(c) for c in a.children], a.type_id, a.byte_width, a.list_size)


def down_nested(d):
    import nested as T
    end = d.offset + d.length
    valid = None
    if d.validity is not None and d.kind != T.K_NULL:
        valid = np.unpackbits(d.validity[:(end + 7) // 8].cpu().numpy(), bitorder="little")[:end].astype(bool)
    buf1 = buf2 = None
    if d.kind in (T.K_PRIM, T.K_DICT):
        raw = d.buf1.cpu().numpy()
        if d.type_id == orc.BOOL:
            buf1 = np.unpackbits(raw, bitorder="little")[:end].astype(bool)
        else:
            buf1 = raw[:end * np.dtype(orc.NP_OF[d.type_id]).itemsize].view(orc.NP_OF[d.type_id]).copy()
    elif d.kind in T.VAR32 + T.VAR64:
        dt = np.int32 if d.kind in T.VAR32 else np.int64
        buf1 = d.buf1[:(end + 1) * np.dtype(dt).itemsize].cpu().numpy().view(dt).copy()
        if d.kind in T.BINARY_KINDS:
            buf2 = d.buf2[:int(buf1[end])].cpu().numpy().copy() if end else np.zeros(0, np.uint8)
    elif d.kind in T.FIXED_KINDS:
        buf1 = d.buf1[:end * d.byte_width].cpu().numpy().copy()
    return T.N(d.kind, d.length, d.offset, valid, d.null_count, buf1, buf2, [down_nested(c) for c in d.children], d.type_id,
               d.byte_width, d.list_size)


class GpuNested:
    def __init__(self, ctx=None):
        self.ctx = ctx or device.Context(0)

    def select(self, values, sel, is_filter, option):
        from arrow1_amd import nested as P
        try:
            if is_filter:
                return down_nested(P.filter(self.ctx, up_nested(values), up(sel), option))
            return down_nested(P.take(self.ctx, up_nested(values), up(sel), bool(option)))
        except _abi.AmdArrowError as e:
            raise orc.StatusError(e.code, e.msg)


    def group(self, batches):
        """batches: list of lists of nested.N key columns -> (ids per batch, num_groups, uniques per column as N)"""
        from arrow1_amd import nested as P
        g = P.Grouper(self.ctx, [(c.kind, c.type_id) for c in batches[0]])
        try:
            ids = []
            for cols in batches:
                try:
                    ids.append(g.consume([up_nested(c) for c in cols]).values_numpy())
                except _abi.AmdArrowError as e:
                    raise orc.StatusError(e.code, e.msg)
            return ids, g.num_groups(), [down_nested(g.uniques(c)) for c in range(len(batches[0]))]
        finally:
            g.close()
