"""Row-range sharded Filter -> GroupBy over the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank runs the
fused kernel on its own contiguous row range -- no communication on the data path -- and the
ONLY exchange is the final group-by merge of the per-rank group tables (<= #groups rows each,
not #rows).  Two merge protocols:

* dense (north_star): integer keys whose global range is small (<= DENSE_LIMIT) index
  identically shaped accumulators on every rank, so the merge is ONE all-reduce(SUM) over
  [sums | counts] and ONE all-reduce(MIN) over a packed (owner rank << 32 | local group id) word
  per key: the minimum names the rank where the key appears first and its position in that
  rank's first-appearance order, which is exactly the global first-appearance order
  (hash_aggregate.cc:391-401) because shards are row ranges in rank order.  ~24 MB on the wire for
  1 M groups instead of world x 20 MB.  The merge runs on a side stream with double-buffered
  outputs, so it overlaps the next batch's fused kernel.
* gather (general keys, nulls in wide keys, huge ranges): all-gather the partial tables and
  re-aggregate them with the same kernel; rank-major concatenation preserves first-appearance order.

int64 sums are exact at any world size; float64 sums carry the usual reordering tolerance.
The protocols are written against torch tensors + torch.distributed so that the world_size-2
`gloo` test on CPU drives exactly this code with a CPU stand-in supplied by the test."""
import torch
import torch.distributed as tdist

DENSE_LIMIT = 1 << 23
_INF = (1 << 62)


class Partial:
    """A group table: keys[G] (+key_valid[G] bool), sums[G] (int64 or float64), counts[G] int64."""

    def __init__(self, keys, key_valid, sums, counts):
        self.keys, self.key_valid, self.sums, self.counts = keys, key_valid, sums, counts

    @property
    def num_groups(self):
        return int(self.keys.shape[0])


class LazyPartial:
    """A Partial that is built when somebody looks at it: a step of the dense protocol needs its local table as
    tensors only if the step has to be redone (the first step's plan, a key outside the plan), and unpacking the key
    validity bitmap of a million groups is four launches of host time between two steps -- the fused call returns only
    when its kernels are all but done, so host work behind it leaves the GPU idle (0.69 -> 0.63 ms per merged step)."""

    def __init__(self, make):
        self._make, self._p = make, None

    def _get(self):
        if self._p is None:
            self._p = self._make()
        return self._p

    keys = property(lambda self: self._get().keys)
    key_valid = property(lambda self: self._get().key_valid)
    sums = property(lambda self: self._get().sums)
    counts = property(lambda self: self._get().counts)
    num_groups = property(lambda self: self._get().num_groups)


# ---- general protocol: all-gather + re-aggregate ---------------------------------------------------
def merge_partials(part, local_merge, world=None, group=None):
    """All-gather the per-rank partial tables and re-aggregate them.

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


# ---- dense protocol: two all-reduces ----------------------------------------------------------------
def key_range(part, group=None):
    """Global [kmin, kmax] of the non-null keys: one 16-byte all-reduce(MIN) over (min, -max)."""
    dev = part.keys.device
    k = part.keys[part.key_valid].to(torch.int64)
    if k.numel():
        mm = torch.stack([k.min(), -k.max()])
    else:
        mm = torch.tensor([_INF, _INF], dtype=torch.int64, device=dev)
    tdist.all_reduce(mm, op=tdist.ReduceOp.MIN, group=group)
    lo, neg_hi = int(mm[0].item()), int(mm[1].item())
    return (lo, -neg_hi) if lo != _INF else (0, -1)


class DenseState:
    """Accumulators reused across steps (allocated once): [sums | counts] and the owner word,
    R + 1 entries (the last one is the null key)."""

    def __init__(self, kmin, R, sum_dtype, device):
        self.kmin, self.R = kmin, R
        self.acc = torch.zeros(2, R + 1, dtype=sum_dtype if sum_dtype == torch.float64 else torch.int64, device=device)
        self.cnt_view = self.acc[1]
        self.owner = torch.full((R + 1,), _INF, dtype=torch.int64, device=device)
        self.bad = torch.zeros(1, dtype=torch.int64, device=device)  # keys seen outside [kmin, kmin+R)


def merge_dense(part, st, rank, group=None):
    """-> (Partial in global first-appearance order, device flag tensor `bad`).  No host sync."""
    dev = part.keys.device
    G = part.num_groups
    st.acc.zero_()
    st.owner.fill_(_INF)
    st.bad.zero_()
    idx = part.keys.to(torch.int64) - st.kmin
    idx = torch.where(part.key_valid, idx, torch.full_like(idx, st.R))
    oob = (idx < 0) | (idx > st.R)
    st.bad += oob.sum()
    idx = idx.clamp(0, st.R)
    is_f = st.acc.dtype == torch.float64
    st.acc[0].index_put_((idx,), part.sums.to(st.acc.dtype))
    st.acc[1].index_put_((idx,), part.counts.to(st.acc.dtype))
    st.owner.index_put_((idx,), (rank << 32) | torch.arange(G, dtype=torch.int64, device=dev))
    tdist.all_reduce(st.acc, op=tdist.ReduceOp.SUM, group=group)     # sums and counts in one message
    tdist.all_reduce(st.owner, op=tdist.ReduceOp.MIN, group=group)   # who saw the key first, and where
    tdist.all_reduce(st.bad, op=tdist.ReduceOp.SUM, group=group)
    # global first-appearance order = ascending (owner rank, local id): one sort of <= R + 1 words
    order = torch.argsort(st.owner)
    n_present = (st.owner != _INF).sum()
    sums = st.acc[0][order]
    counts = st.acc[1][order]
    keys = (order + st.kmin)
    key_valid = order != st.R
    return Partial(keys, key_valid, sums, counts if not is_f else counts.to(torch.int64)), n_present


# ---- which protocol, and what happens when the data leave the plan ---------------------------------------------
class ShardedMerge:
    """The protocol state machine of the sharded group-by, independent of where the tables live (HBM + RCCL in
    GroupByMerger, torch CPU tensors + gloo in the tests).

    The first step plans: one 16-byte all-reduce gives the global key range; a small integer range takes the dense
    protocol (two all-reduces per step into accumulators laid out once, with head-room), anything else the gather
    protocol.  A later step whose keys leave the dense range is NOT an error: every rank sees the same out-of-range
    count (it travels inside the SUM all-reduce), so at wait() all ranks together redo THAT step with the gather
    protocol -- the step's local table is still there -- and the next step re-plans over the union of the old range
    and the range that step brought.  All collectives are issued in program order on every rank.

    make_state(kmin, R) -> state for one slot; dense_step(part, state) -> (Partial, n_present, bad) with `bad` a
    one-element tensor every rank agrees on; gather_step(part) -> Partial."""

    def __init__(self, world, make_state, dense_step, gather_step, group=None, integer_keys=True, slots=2):
        self.world, self.group, self.integer_keys, self.slots = world, group, integer_keys, slots
        self.make_state, self.dense_step, self.gather_step = make_state, dense_step, gather_step
        self.plan = None      # None: not planned; False: gather protocol; (kmin, R): dense protocol
        self.states = None
        self.seen = None      # the key range every step so far has shown (the union the next plan must cover)
        self.replans = 0

    def _make_plan(self, part):
        lo, hi = key_range(part, self.group)
        if self.seen is not None and hi >= lo:
            lo, hi = min(lo, self.seen[0]), max(hi, self.seen[1])
        elif self.seen is not None:
            lo, hi = self.seen
        if hi >= lo:
            self.seen = (lo, hi)
        R = hi - lo + 1
        if self.integer_keys and 0 < R <= DENSE_LIMIT:
            pad = max(1024, R // 8)  # head-room: later batches may see keys a little outside the range so far
            self.plan = (lo - pad, R + 2 * pad)
            self.states = [self.make_state(*self.plan) for _ in range(self.slots)]
        else:
            self.plan, self.states = False, None

    def step(self, part, slot=0, run_dense=None):
        """-> PendingMerge.  `run_dense(state)` (optional) replaces dense_step(part, state): GroupByMerger queues the
        device merge on its side stream there."""
        if self.plan is None:
            self._make_plan(part)
        if self.plan is False:
            return PendingMerge(self, part, self.gather_step(part), None, None)
        merged, n_present, bad = (run_dense or (lambda st: self.dense_step(part, st)))(self.states[slot])
        return PendingMerge(self, part, merged, n_present, bad)

    def _left_the_plan(self, part):
        """Collective (every rank is here, in the same step): the step again on the gather protocol, re-plan next time."""
        self.replans += 1
        lo, hi = key_range(part, self.group)
        if hi >= lo:
            self.seen = (min(lo, self.seen[0]), max(hi, self.seen[1])) if self.seen else (lo, hi)
        self.plan, self.states = None, None
        return self.gather_step(part)


class PendingMerge:
    def __init__(self, owner, part, merged, n_present, bad, event=None):
        self.owner, self.part, self.merged, self.n_present, self.bad, self.event = owner, part, merged, n_present, bad, event
        self.bad_host = None  # (GroupByMerger: `bad` copied to pinned host memory on the merge's stream, in front of `event`)
        self.done = False

    def wait(self):
        if self.done:
            return self
        if self.event is not None:
            self.event.synchronize()
        if self.bad_host is not None:  # no second round trip to the device: the flag arrived with the event
            left = float(self.bad_host[0]) != 0
        else:
            left = self.bad is not None and int(self.bad.item()) != 0
        if left:
            self.merged, self.n_present = self.owner._left_the_plan(self.part), None
        self.done = True
        return self

    @property
    def num_groups(self):
        self.wait()
        if self.n_present is None:
            return self.merged.num_groups
        return int(self.n_present.item()) if torch.is_tensor(self.n_present) else int(self.n_present)

    def tensors(self):
        G = self.num_groups
        p = self.merged
        return p.keys[:G], p.key_valid[:G], p.sums[:G], p.counts[:G]


# ---- dense protocol on the device: the HIP kernels of csrc/dist_merge.hip around the two all-reduces ----
class DenseDeviceState:
    """Accumulators and outputs of the device merge, allocated once per plan: acc[R + 2][2]
    ({sum, count} per key; column R = the null key, acc[R + 1][0] = keys outside the range), owner[R + 1],
    and the merged table (capacity rows) with its group count in a device word."""

    def __init__(self, kmin, R, is_float, key_torch, capacity, device):
        self.kmin, self.R, self.is_float, self.capacity = kmin, R, is_float, capacity
        self.acc = torch.zeros(R + 2, 2, dtype=torch.float64 if is_float else torch.int64, device=device)
        self.owner = torch.zeros(R + 2, dtype=torch.int64, device=device)  # [R + 1]: -(largest local table) after the MIN
        rows = R + 1
        self.rows = rows
        self.out_keys = torch.zeros(rows, dtype=key_torch, device=device)
        self.out_valid = torch.zeros(rows, dtype=torch.uint8, device=device)
        self.out_sums = torch.zeros(rows, dtype=torch.float64 if is_float else torch.int64, device=device)
        self.out_counts = torch.zeros(rows, dtype=torch.int64, device=device)
        self.n_present = torch.zeros(1, dtype=torch.int64, device=device)

    @property
    def bad(self):
        return self.acc[self.R + 1, 0:1]


def merge_dense_device(ctx, key_type, r_keys, r_key_validity, r_sums, r_counts, num_groups, st, rank, world,
                       all_reduce=None):
    """One rank's merge step on the device.  r_*: the fused kernel's output buffers (keys, validity
    BITMAP, 8-byte sums, int64 counts; uint8 tensors).  all_reduce(tensor, op): defaults to
    torch.distributed (RCCL); the single-GPU test passes a stand-in that combines simulated ranks.
    -> (Partial over the preallocated outputs, n_present device tensor).  No host sync."""
    from ._abi import check
    lib, h = ctx.lib, ctx.handle
    check(lib, lib.amdarrow_dense_merge_scatter(h, r_keys.data_ptr(), key_type, r_key_validity.data_ptr(), 1, r_sums.data_ptr(),
                                                r_counts.data_ptr(), num_groups, st.kmin, st.R, rank, int(st.is_float),
                                                st.acc.data_ptr(), st.owner.data_ptr()))
    if all_reduce is None:
        tdist.all_reduce(st.acc, op=tdist.ReduceOp.SUM)     # sums, counts and the out-of-range count in one message
        tdist.all_reduce(st.owner, op=tdist.ReduceOp.MIN)   # who saw the key first, and where in its order
    else:
        all_reduce(st.acc, "sum")
        all_reduce(st.owner, "min")
    check(lib, lib.amdarrow_dense_merge_order(h, st.acc.data_ptr(), st.owner.data_ptr(), st.kmin, st.R, world, st.capacity, key_type,
                                              int(st.is_float), st.rows, st.out_keys.data_ptr(), st.out_valid.data_ptr(),
                                              st.out_sums.data_ptr(), st.out_counts.data_ptr(), st.n_present.data_ptr()))
    return Partial(st.out_keys, st.out_valid.view(torch.bool), st.out_sums, st.out_counts), st.n_present


# ---- sharded sort_indices: a sample sort over the row-range shards ------------------
# sort_indices does not shard with one all-reduce: the rows have to move.  Every rank sorts its own row
# range (stable, NaNs and nulls last: the single-GPU kernel), the ranks agree on world - 1 splitters from
# evenly spaced samples of their sorted keys, every rank cuts its sorted run at the splitters and sends
# piece j to rank j (all-to-all; NaN and null rows go to the last rank), and rank j sorts what it
# received -- the runs arrive in source-rank order, each in local index order among equal keys, so the
# stable local sort yields the global order of vector_sort.cc:373-575 including its tie-breaking by
# index.  Rank j ends up with the global row indices of the j-th key range; the concatenation over the
# ranks is sort_indices of the whole column (a ChunkedArray with `world` chunks).
# `ops` binds the array operation to the HIP kernels (SortOps below) or, in the CPU / gloo test, to a
# CPU stand-in the test supplies: argsort(values, valid, descending) -> int64 permutation.
_UNSIGNED_MASK = {"uint8": 0xFF, "uint16": 0xFFFF, "uint32": 0xFFFFFFFF}


def _ordered(t, kind, descending):
    """A few gathered keys -> a tensor whose `<` is the sort order (float64, or int64 with the order kept)."""
    if kind in ("float32", "float64"):
        v = t.to(torch.float64)
        return -v if descending else v
    if kind == "uint64":
        v = t.view(torch.int64) ^ (-2 ** 63)  # order-preserving map of the unsigned range onto int64
    elif kind in _UNSIGNED_MASK:
        v = t.view({"uint8": torch.int8, "uint16": torch.int16, "uint32": torch.int32}[kind]).to(torch.int64) & _UNSIGNED_MASK[kind]
    else:
        v = t.to(torch.int64)
    return ~v if descending else v  # ~x reverses the order of two's complement integers without overflow


def _exchange(send, send_counts, recv_counts, group=None):
    """Variable all-to-all of rows of `send` (first dimension), pieces laid out in destination order."""
    world = len(send_counts)
    out = torch.empty((int(sum(recv_counts)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    if tdist.get_backend(group) == "nccl":
        tdist.all_to_all_single(out, send.contiguous(), [int(c) for c in recv_counts], [int(c) for c in send_counts], group=group)
        return out
    # gloo has no all-to-all: gather every rank's padded send buffer and cut out what is addressed here
    rank = tdist.get_rank(group)
    dev = send.device
    cap = torch.tensor([send.shape[0]], dtype=torch.int64, device=dev)
    caps = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    tdist.all_gather(caps, cap, group=group)
    m = int(max(int(c.item()) for c in caps))
    padded = torch.zeros((max(m, 1),) + tuple(send.shape[1:]), dtype=send.dtype, device=dev)
    padded[:send.shape[0]] = send
    bufs = [torch.zeros_like(padded) for _ in range(world)]
    tdist.all_gather(bufs, padded, group=group)
    sc = torch.tensor([int(c) for c in send_counts], dtype=torch.int64, device=dev)
    all_sc = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    tdist.all_gather(all_sc, sc, group=group)
    pos = 0
    for src in range(world):
        counts = [int(x) for x in all_sc[src].tolist()]
        start = sum(counts[:rank])
        out[pos:pos + counts[rank]] = bufs[src][start:start + counts[rank]]
        pos += counts[rank]
    return out


def sharded_sort_indices(ops, values, valid, row0, kind, descending, rank, world, group=None, samples=64):
    """values: this rank's rows (1-D tensor; uint64 as an int64 view), valid: bool tensor or None, row0: global
    index of its first row, kind: numpy-style dtype name.  -> int64 tensor: global row indices of this
    rank's key range, in sorted order."""
    dev = values.device
    n = values.shape[0]
    is_float = kind in ("float32", "float64")
    perm = ops.argsort(values, valid, descending)
    sv = values[perm]
    svalid = valid[perm] if valid is not None else torch.ones(n, dtype=torch.bool, device=dev)
    n_null = int((~svalid).sum().item())
    n_nan = int(torch.isnan(sv[:n - n_null]).sum().item()) if is_float and n > n_null else 0
    n0 = n - n_null - n_nan  # [0, n0) sortable keys in order, then the NaNs, then the nulls
    # --- splitters from evenly spaced samples of every rank's sorted keys
    take = torch.linspace(0, max(n0 - 1, 0), samples, device=dev).to(torch.int64) if n0 > 0 else torch.zeros(0, dtype=torch.int64, device=dev)
    mine = _ordered(sv[take], kind, descending) if n0 > 0 else torch.zeros(0, dtype=torch.float64 if is_float else torch.int64, device=dev)
    padded = torch.zeros(samples, dtype=mine.dtype, device=dev)
    padded[:mine.shape[0]] = mine
    have = torch.tensor([mine.shape[0]], dtype=torch.int64, device=dev)
    all_s = [torch.zeros_like(padded) for _ in range(world)]
    all_h = [torch.zeros_like(have) for _ in range(world)]
    tdist.all_gather(all_s, padded, group=group)
    tdist.all_gather(all_h, have, group=group)
    pool = torch.cat([all_s[r][:int(all_h[r].item())] for r in range(world)])
    pool = torch.sort(pool).values
    if pool.shape[0] > 0:
        at = (torch.arange(1, world, device=dev) * pool.shape[0]) // world
        splitters = pool[at.clamp(max=pool.shape[0] - 1)]
    else:
        splitters = torch.zeros(world - 1, dtype=padded.dtype, device=dev)
    # --- cut the sorted run: piece j = keys k with splitter[j-1] < k <= splitter[j] (upper bounds by bisection,
    # on the device, a handful of gathered keys per step: no pass over the data, no dtype restrictions)
    lo = torch.zeros(world - 1, dtype=torch.int64, device=dev)
    hi = torch.full((world - 1,), n0, dtype=torch.int64, device=dev)
    for _ in range(max(1, int(n0).bit_length())):
        mid = (lo + hi) // 2
        probe = _ordered(sv[mid.clamp(max=max(n0 - 1, 0))], kind, descending) if n0 > 0 else splitters
        go_right = (probe <= splitters) & (mid < hi)
        lo = torch.where(go_right, mid + 1, lo)
        hi = torch.where(go_right, hi, mid)
    bounds = [0] + [int(x) for x in lo.tolist()] + [n0]
    send_counts = [bounds[j + 1] - bounds[j] for j in range(world)]
    send_counts[-1] += n_nan + n_null  # NaNs and nulls sort after every key: they all belong to the last rank
    counts_t = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    all_c = [torch.zeros_like(counts_t) for _ in range(world)]
    tdist.all_gather(all_c, counts_t, group=group)
    recv_counts = [int(all_c[src][rank].item()) for src in range(world)]
    # --- move keys, validity and global row numbers
    gidx = perm.to(torch.int64) + row0
    rv = _exchange(sv, send_counts, recv_counts, group)
    rvalid = _exchange(svalid.to(torch.uint8), send_counts, recv_counts, group).to(torch.bool)
    ridx = _exchange(gidx, send_counts, recv_counts, group)
    # --- the received runs are in source-rank order: one more stable sort finishes this rank's range
    perm2 = ops.argsort(rv, None if bool(rvalid.all()) else rvalid, descending)
    return ridx[perm2]


class SortOps:
    """sharded_sort_indices' array operation on the HIP kernels."""

    def __init__(self, ctx, type_id):
        self.ctx, self.type_id = ctx, type_id

    def argsort(self, values, valid, descending):
        from . import _abi, compute
        from .device import DeviceArray
        n = values.shape[0]
        if n == 0:
            return torch.zeros(0, dtype=torch.int64, device=values.device)
        vbuf = None if valid is None else _pack_bits(valid)
        arr = DeviceArray(self.type_id, values.contiguous().view(torch.uint8), vbuf, n, 0, -1)
        r = compute.sort_indices(self.ctx, arr, _abi.DESCENDING if descending else _abi.ASCENDING)
        return r.values[:n * 8].view(torch.int64)


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
        self.ctx, self.rank, self.world = ctx, rank, world
        self.key_type, self.value_type, self.cap = key_type, value_type, group_capacity
        self.is_float = value_type in (_abi.FLOAT, _abi.DOUBLE)
        self.sum_type = _abi.DOUBLE if self.is_float else _abi.INT64
        self.key_torch = {_abi.INT8: torch.int8, _abi.UINT8: torch.uint8, _abi.INT16: torch.int16,
                          _abi.INT32: torch.int32, _abi.INT64: torch.int64}.get(key_type)
        if self.key_torch is None:
            raise NotImplementedError("sharded group-by: key type %d" % key_type)
        self._merge_bufs = None
        self._sm = ShardedMerge(world, self._make_state, None, self._gather_step)
        self._side = torch.cuda.Stream(device=ctx.device) if world > 1 else None
        self._side_ctx = None
        self._slot = 0
        self._bufs = None
        self._pending = [None, None]
        self._bad_host, self._bad_used = None, None

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

    def _gather_step(self, part):
        """the gather protocol for one step: all-gather + re-aggregation on the device; the table is copied out of the
        merge buffers (the next fallback reuses them)"""
        p = self._partial_of(merge_partials(part, self._local_merge, self.world))
        return Partial(p.keys.clone(), p.key_valid.clone(), p.sums.clone(), p.counts.clone())

    def _make_state(self, kmin, R):
        return DenseDeviceState(kmin, R, self.is_float, self.key_torch, self.cap, self.ctx.device)

    @property
    def replans(self):
        return self._sm.replans

    def filter_groupby(self, values, keys, mask, buffers=None):
        """One step.  world == 1: the fused kernel's own result.  world > 1: a PendingMerge whose merge runs on a side
        stream (dense protocol) and overlaps the caller's next step; a step whose keys leave the planned dense range is
        redone on the gather protocol when it is waited for, and the plan is widened (ShardedMerge)."""
        from . import compute
        if self.world == 1:
            return compute.filter_groupby(self.ctx, values, keys, mask, buffers=buffers)
        if self._bufs is None:
            self._bufs = [compute.GroupByBuffers(self.ctx.device, self.cap, self.key_type) for _ in range(2)]
        slot = self._slot
        self._slot ^= 1
        if self._pending[slot] is not None:   # the buffers of step i-2 are about to be overwritten
            self._pending[slot].wait()
        r = compute.filter_groupby(self.ctx, values, keys, mask, buffers=self._bufs[slot])

        def run_dense(state):
            if self._side_ctx is None:  # the merge kernels run on the side stream: a context of their own, bound to it
                from .device import Context
                with torch.cuda.stream(self._side):
                    self._side_ctx = Context(self.ctx.device.index)
            main = torch.cuda.current_stream(self.ctx.device)
            done_local = torch.cuda.Event()
            done_local.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(done_local)
                mpart, n_present = merge_dense_device(self._side_ctx, self.key_type, r.keys, r.key_validity, r.sums, r.counts, r.num_groups,
                                                      state, self.rank, self.world)
                if self._bad_host is None:
                    self._bad_host = [torch.zeros(1, dtype=state.bad.dtype).pin_memory() for _ in range(2)]
                self._bad_host[slot].copy_(state.bad, non_blocking=True)
                self._bad_used = self._bad_host[slot]
                self._event = torch.cuda.Event()
                self._event.record(self._side)
            return mpart, n_present, state.bad

        self._event, self._bad_used = None, None
        res = self._sm.step(LazyPartial(lambda: self._partial_of(r)), slot, run_dense)
        res.event = self._event
        res.bad_host = self._bad_used
        self._pending[slot] = res
        return res

    def finish(self):
        """Wait for the merges still in flight (call before reading the clock)."""
        for p in self._pending:
            if p is not None:
                p.wait()
