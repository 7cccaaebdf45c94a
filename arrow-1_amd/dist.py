"""Row-range sharded Filter -> GroupBy over the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank runs the
fused kernel on its own contiguous row range -- no communication on the data path -- and the
ONLY exchange is the final group-by merge: the per-rank group tables (<= #groups rows each,
not #rows) are all-gathered and re-aggregated with the same kernel.  Because shards are row
ranges in rank order and every partial table is in local first-appearance order, the
first-appearance order of the concatenation (rank-major) IS the global first-appearance order
(hash_aggregate.cc:391-401), so the merged result is row-for-row what a single GroupBy over
all rows returns; int64 sums are exact at any world size, float64 sums carry the usual
reordering tolerance.

The protocol is written against torch tensors and two callables so that the world_size-2
`gloo` test on CPU (tests/test_dist_gloo.py) can drive exactly this protocol with a CPU
stand-in supplied BY THE TEST for the two kernel callables."""
import torch
import torch.distributed as tdist


class Partial:
    """A group table: keys[G] (+key_valid[G] bool), sums[G] (int64 or float64), counts[G] int64."""

    def __init__(self, keys, key_valid, sums, counts):
        self.keys, self.key_valid, self.sums, self.counts = keys, key_valid, sums, counts

    @property
    def num_groups(self):
        return int(self.keys.shape[0])


def merge_partials(part, local_merge, world=None, group=None):
    """All-gather the per-rank partial tables and re-aggregate them.

    part: this rank's Partial (tensors on the communicating device).
    local_merge(keys, key_valid, sums, counts, live) -> Partial: a GroupBy over the gathered rows
    where `live` (bool) marks real (non-padding) rows; sums and counts are both summed per key."""
    world = tdist.get_world_size(group) if world is None else world
    dev = part.keys.device
    g_local = torch.tensor([part.num_groups], dtype=torch.int64, device=dev)
    g_all = torch.empty(world, dtype=torch.int64, device=dev)
    tdist.all_gather_into_tensor(g_all, g_local, group=group)   # 8 B per rank
    gmax = int(g_all.max().item())
    if gmax == 0:
        return part

    def pad(t):
        out = torch.zeros(gmax, dtype=t.dtype, device=dev)
        out[:t.shape[0]] = t
        return out

    def gather(t):
        out = torch.empty(world * gmax, dtype=t.dtype, device=dev)
        tdist.all_gather_into_tensor(out, pad(t), group=group)
        return out

    keys = gather(part.keys)
    key_valid = gather(part.key_valid.to(torch.uint8)).to(torch.bool)
    sums = gather(part.sums)
    counts = gather(part.counts)
    slot = torch.arange(gmax, device=dev).repeat(world)
    live = slot < g_all.repeat_interleave(gmax)
    return local_merge(keys, key_valid, sums, counts, live)


# ---- GPU binding -------------------------------------------------------------------------------
def _pack_bits(bits):
    """bool tensor -> Arrow LSB-first bitmap (uint8 tensor, padded)"""
    from .device import alloc_bytes
    n = bits.shape[0]
    nb = (n + 7) // 8
    padded = torch.zeros(nb * 8, dtype=torch.uint8, device=bits.device)
    padded[:n] = bits.to(torch.uint8)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=bits.device)
    packed = (padded.view(-1, 8) * w).sum(1, dtype=torch.int32).to(torch.uint8)
    buf = alloc_bytes(nb, bits.device, zero=True)
    buf[:nb] = packed
    return buf


def _unpack_bits(buf, n):
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=buf.device)
    return ((buf[:(n + 7) // 8].unsqueeze(1) & w) != 0).reshape(-1)[:n]


class GroupByMerger:
    """Sharded Filter -> GroupBy{hash_sum, hash_count} on the HIP kernels."""

    def __init__(self, ctx, rank, world, key_type, value_type, group_capacity):
        from . import _abi
        from .device import NP_OF
        self.ctx, self.rank, self.world = ctx, rank, world
        self.key_type, self.value_type, self.cap = key_type, value_type, group_capacity
        self.is_float = value_type in (_abi.FLOAT, _abi.DOUBLE)
        self.sum_type = _abi.DOUBLE if self.is_float else _abi.INT64
        self.key_torch = {_abi.INT8: torch.int8, _abi.UINT8: torch.uint8, _abi.INT16: torch.int16,
                          _abi.INT32: torch.int32, _abi.INT64: torch.int64}.get(key_type)
        if self.key_torch is None:
            raise NotImplementedError("sharded group-by: key type %d" % key_type)
        self._merge_bufs = None

    def _partial_of(self, r):
        G = r.num_groups
        kw = torch.tensor([], dtype=self.key_torch).element_size()
        keys = r.keys[:G * kw].view(self.key_torch)
        sums = r.sums[:G * 8].view(torch.float64 if self.is_float else torch.int64)
        counts = r.counts[:G * 8].view(torch.int64)
        return Partial(keys, _unpack_bits(r.key_validity, G), sums, counts)

    def _local_merge(self, keys, key_valid, sums, counts, live):
        from . import _abi, compute
        from .device import DeviceArray
        n = keys.shape[0]
        kv = None if bool(key_valid.all()) else _pack_bits(key_valid)
        karr = DeviceArray(self.key_type, keys.contiguous().view(torch.uint8), kv, n, 0, -1)
        marr = DeviceArray(_abi.BOOL, _pack_bits(live), None, n)
        if self._merge_bufs is None:
            self._merge_bufs = (compute.GroupByBuffers(keys.device, self.cap, self.key_type),
                                compute.GroupByBuffers(keys.device, self.cap, self.key_type))
        b1, b2 = self._merge_bufs
        sarr = DeviceArray(self.sum_type, sums.contiguous().view(torch.uint8), None, n)
        carr = DeviceArray(_abi.INT64, counts.contiguous().view(torch.uint8), None, n)
        rs = compute.filter_groupby(self.ctx, sarr, karr, marr, buffers=b1)   # sum of partial sums
        rc = compute.filter_groupby(self.ctx, carr, karr, marr, buffers=b2)   # sum of partial counts
        rs.counts = rc.sums  # merged valid-value counts (int64)
        return rs

    def filter_groupby(self, values, keys, mask, buffers):
        from . import compute
        r = compute.filter_groupby(self.ctx, values, keys, mask, buffers=buffers)
        if self.world == 1:
            return r
        return merge_partials(self._partial_of(r), self._local_merge, self.world)
