"""Host-side mirror (Python flavour) of the reference's operator API for the hot path:
same function names and option meaning as arrow::compute::CallFunction
(compute/exec.h:229-237, api_vector.h, api_scalar.h, api_aggregate.h), every call going
straight to the HIP kernels through the C ABI (include/amdarrow.h).  The C++ flavour --
what an Arrow C++ caller links -- lives in arrow-1_amd/host/.  Nothing here computes on
the CPU."""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._abi import (BOOL, UINT8, INT8, UINT16, INT16, UINT32, INT32, UINT64, INT64, FLOAT, DOUBLE, DROP, EMIT_NULL, ASCENDING, DESCENDING, COut, CArray, AmdArrowError, check)
from .device import Context, DeviceArray, alloc_bytes, bit_width, NP_OF

_CMP = {"equal": _abi.EQ, "not_equal": _abi.NE, "greater": _abi.GT, "greater_equal": _abi.GE,
        "less": _abi.LT, "less_equal": _abi.LE}
_ARITH = {"add": _abi.ADD, "subtract": _abi.SUB, "multiply": _abi.MUL, "divide": _abi.DIV,
          "add_checked": _abi.ADD_CHECKED, "subtract_checked": _abi.SUB_CHECKED,
          "multiply_checked": _abi.MUL_CHECKED, "divide_checked": _abi.DIV_CHECKED,
          "power": _abi.POWER, "power_checked": _abi.POWER_CHECKED, "negate": _abi.NEGATE, "negate_checked": _abi.NEGATE_CHECKED}
_AGG = {"hash_sum": _abi.HASH_SUM, "hash_count": _abi.HASH_COUNT, "hash_mean": _abi.HASH_MEAN}


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _values_bytes(type_id, n):
    return (n + 7) // 8 if type_id == BOOL else n * bit_width(type_id) // 8


def filter(ctx, values, mask, null_selection=DROP, out_capacity=None):  # noqa: A001 (reference name)
    """CallFunction("filter", {values, mask}, FilterOptions) -- vector_selection.cc:786-835."""
    cap = values.length if out_capacity is None else out_capacity
    dev = values.values.device
    out_v = alloc_bytes(_values_bytes(values.type_id, cap), dev)
    out_b = alloc_bytes((cap + 7) // 8, dev)
    out = COut()
    cv, cm = values.c(), mask.c()
    check(ctx.lib, ctx.lib.amdarrow_filter(ctx.handle, C.byref(cv), C.byref(cm), null_selection, _ptr(out_v),
                                           _ptr(out_b), cap, C.byref(out)))
    return DeviceArray(values.type_id, out_v, out_b if out.has_validity else None, out.length, 0, out.null_count)


def filter_output_size(ctx, mask, null_selection=DROP):
    n = C.c_int64(0)
    cm = mask.c()
    check(ctx.lib, ctx.lib.amdarrow_filter_output_size(ctx.handle, C.byref(cm), null_selection, C.byref(n)))
    return n.value


def take_indices(ctx, mask, null_selection=DROP):
    """internal::GetTakeIndices -- vector_selection.cc:92-237."""
    cap = mask.length
    dev = mask.values.device
    out_v = alloc_bytes(cap * 4, dev)
    out_b = alloc_bytes((cap + 7) // 8, dev)
    out, w = COut(), C.c_int(0)
    cm = mask.c()
    check(ctx.lib, ctx.lib.amdarrow_take_indices(ctx.handle, C.byref(cm), null_selection, _ptr(out_v), _ptr(out_b),
                                                 cap, C.byref(w), C.byref(out)))
    t = _abi.UINT16 if w.value == 16 else _abi.UINT32
    return DeviceArray(t, out_v, out_b if out.has_validity else None, out.length, 0, out.null_count)


def take(ctx, values, indices, boundscheck=True):
    """CallFunction("take", {values, indices}, TakeOptions) -- vector_selection.cc:493-530."""
    n = indices.length
    dev = values.values.device
    out_v = alloc_bytes(_values_bytes(values.type_id, n), dev)
    out_b = alloc_bytes((n + 7) // 8, dev)
    out = COut()
    cv, ci = values.c(), indices.c()
    check(ctx.lib, ctx.lib.amdarrow_take(ctx.handle, C.byref(cv), C.byref(ci), int(boundscheck), _ptr(out_v),
                                         _ptr(out_b), C.byref(out)))
    return DeviceArray(values.type_id, out_v, out_b, out.length, 0, out.null_count)


def binary_take(ctx, values, indices, boundscheck=True):
    """CallFunction("take") on a BinaryArray / StringArray -- VarBinaryImpl, vector_selection.cc:1493-1573."""
    from .device import DeviceBinaryArray
    n = indices.length
    dev = values.offsets.device
    out_o = alloc_bytes(4 * (n + 1), dev)
    out_b = alloc_bytes((n + 63) // 64 * 8, dev)
    out, nbytes = COut(), C.c_int64(0)
    cv, ci = values.c(), indices.c()
    check(ctx.lib, ctx.lib.amdarrow_binary_take_offsets(ctx.handle, C.byref(cv), C.byref(ci), int(boundscheck), _ptr(out_o), _ptr(out_b),
                                                        C.byref(nbytes), C.byref(out)))
    out_d = alloc_bytes(nbytes.value, dev)
    check(ctx.lib, ctx.lib.amdarrow_binary_take_data(ctx.handle, C.byref(cv), C.byref(ci), _ptr(out_o), _ptr(out_d)))
    return DeviceBinaryArray(out_o, out_d, out_b, out.length, 0, out.null_count)


def binary_concatenate(ctx, chunks):
    """arrow::Concatenate of BinaryArrays / StringArrays (array/concatenate.cc:100-160, 203-214)."""
    from ._abi import CBinaryArray
    from .device import DeviceBinaryArray
    arr = (CBinaryArray * len(chunks))(*[c.c() for c in chunks])
    rows, nbytes = C.c_int64(0), C.c_int64(0)
    check(ctx.lib, ctx.lib.amdarrow_binary_concatenate_size(ctx.handle, arr, len(chunks), C.byref(rows), C.byref(nbytes)))
    dev = chunks[0].offsets.device
    out_o = alloc_bytes(4 * (rows.value + 1), dev)
    out_d = alloc_bytes(nbytes.value, dev)
    out_b = alloc_bytes((rows.value + 63) // 64 * 8, dev)
    out = COut()
    check(ctx.lib, ctx.lib.amdarrow_binary_concatenate(ctx.handle, arr, len(chunks), _ptr(out_o), _ptr(out_d), _ptr(out_b), C.byref(out)))
    return DeviceBinaryArray(out_o, out_d, out_b if out.has_validity else None, out.length, 0, out.null_count)


def binary_filter(ctx, values, mask, null_selection=DROP):
    """CallFunction("filter") on a BinaryArray / StringArray -- BinaryFilter, vector_selection.cc:840-1124: the Take of
    GetTakeIndices(mask), with BinaryFilter's conventions for the validity buffer and the null count (:1090-1112)."""
    from .device import DeviceBinaryArray
    idx = take_indices(ctx, mask, null_selection)
    r = binary_take(ctx, values, idx, boundscheck=False)
    vnc = values.null_count if values.validity is not None else 0
    fnc = mask.null_count if mask.validity is not None else 0
    keep_validity = vnc != 0 or fnc != 0
    null_count = 0 if (vnc == 0 and (null_selection == DROP or fnc == 0)) else -1
    return DeviceBinaryArray(r.offsets, r.data, r.validity if keep_validity else None, r.length, 0, null_count)


def sort_indices(ctx, chunks, order=ASCENDING):
    """CallFunction("sort_indices", {array | chunked_array}, SortOptions) -- vector_sort.cc:1565-1630."""
    if isinstance(chunks, DeviceArray):
        chunks = [chunks]
    total = sum(c.length for c in chunks)
    dev = chunks[0].values.device if chunks else "cuda"
    out = alloc_bytes(total * 8, dev)
    arr = (CArray * max(len(chunks), 1))(*[c.c() for c in chunks])
    check(ctx.lib, ctx.lib.amdarrow_sort_indices(ctx.handle, arr, len(chunks), order, _ptr(out)))
    return DeviceArray(_abi.UINT64, out, None, total, 0, 0)


def sort_indices_multi(ctx, keys, orders):
    """CallFunction("sort_indices", {record_batch | table}, SortOptions{sort_keys}) --
    vector_sort.cc:1632-1707.  keys: one list of chunks (DeviceArray) per sort key."""
    keys = [[k] if isinstance(k, DeviceArray) else list(k) for k in keys]
    flat = [c for k in keys for c in k]
    total = sum(c.length for c in keys[0]) if keys else 0
    dev = flat[0].values.device if flat else "cuda"
    out = alloc_bytes(total * 8, dev)
    arr = (CArray * max(len(flat), 1))(*[c.c() for c in flat])
    nch = (C.c_int * max(len(keys), 1))(*[len(k) for k in keys])
    ords = (C.c_int * max(len(keys), 1))(*[int(o) for o in orders])
    check(ctx.lib, ctx.lib.amdarrow_sort_indices_multi(ctx.handle, arr, nch, ords, len(keys), _ptr(out)))
    return DeviceArray(_abi.UINT64, out, None, total, 0, 0)


def concatenate(ctx, chunks):
    """arrow::Concatenate (array/concatenate.cc:171-201) of fixed-width / boolean arrays."""
    total = sum(c.length for c in chunks)
    t = chunks[0].type_id
    dev = chunks[0].values.device
    out_v = alloc_bytes((_values_bytes(t, total) + 7) // 8 * 8, dev)
    out_b = alloc_bytes((total + 63) // 64 * 8, dev)
    out = COut()
    arr = (CArray * len(chunks))(*[c.c() for c in chunks])
    check(ctx.lib, ctx.lib.amdarrow_concatenate(ctx.handle, arr, len(chunks), _ptr(out_v), _ptr(out_b), C.byref(out)))
    return DeviceArray(t, out_v, out_b if out.has_validity else None, out.length, 0, out.null_count)


def _binary(ctx, fn, op, lhs, rhs, out_type):
    arr = lhs if isinstance(lhs, DeviceArray) else rhs
    t, n = arr.type_id, arr.length
    keep = []

    def operand(x):
        if isinstance(x, DeviceArray):
            c = x.c()
            keep.append(c)
            return C.byref(c), None
        s = np.array([x], dtype=np.uint8 if t == BOOL else NP_OF[t])
        keep.append(s)
        return None, s.ctypes.data_as(C.c_void_p)

    la, ls = operand(lhs)
    ra, rs = operand(rhs)
    ot = t if out_type is None else out_type
    dev = arr.values.device
    out_v = alloc_bytes(_values_bytes(ot, n), dev)
    out_b = alloc_bytes((n + 7) // 8, dev)
    out = COut()
    check(ctx.lib, fn(ctx.handle, op, la, ls, ra, rs, _ptr(out_v), _ptr(out_b), C.byref(out)))
    return DeviceArray(ot, out_v, out_b, out.length, 0, out.null_count)


_INT_BITS = {UINT8: 8, INT8: 8, UINT16: 16, INT16: 16, UINT32: 32, INT32: 32, UINT64: 64, INT64: 64}
_SIGNED = (INT8, INT16, INT32, INT64)


def common_numeric(type_a, type_b):
    """CommonNumeric (codegen_internal.cc:220-268): the type both operands of a binary compare / arithmetic
    call are cast to when no kernel matches exactly; None when one of them is not a number."""
    ts = (type_a, type_b)
    if any(t not in _INT_BITS and t not in (FLOAT, DOUBLE) for t in ts):
        return None
    if DOUBLE in ts:
        return DOUBLE
    if FLOAT in ts:
        return FLOAT
    ws = max([_INT_BITS[t] for t in ts if t in _SIGNED], default=0)
    wu = max([_INT_BITS[t] for t in ts if t not in _SIGNED], default=0)
    if ws == 0:
        return {64: UINT64, 32: UINT32, 16: UINT16}.get(wu, INT8)  # sic (:252)
    if ws <= wu:
        ws = 1 << wu.bit_length()  # NextPower2(wu + 1)
    return INT64 if ws >= 64 else {32: INT32, 16: INT16}.get(ws, INT8)


def cast(ctx, a, to_type, safe=True):
    """Number -> number cast (scalar_cast_numeric.cc:38-50, 203-262; CastOptions::Safe unless safe=False).
    The result shares the input's validity buffer (and keeps its offset into it)."""
    dev = a.values.device
    out_v = alloc_bytes(_values_bytes(to_type, a.length), dev)
    c = a.c()
    check(ctx.lib, ctx.lib.amdarrow_cast_numeric(ctx.handle, C.byref(c), int(to_type), int(bool(safe)), _ptr(out_v)))
    if a.validity is None:
        return DeviceArray(to_type, out_v, None, a.length, 0, 0)
    if a.offset % 8 == 0:
        return DeviceArray(to_type, out_v, a.validity[a.offset // 8:], a.length, 0, a.null_count)
    # a validity view that does not start on a byte: values re-based in front of element 0 is not expressible with
    # one buffer per array here, so the bits are normalised to offset 0 (is_valid gives exactly that bitmap)
    bits = is_valid(ctx, a)
    return DeviceArray(to_type, out_v, bits.values, a.length, 0, a.null_count)


def _implicitly_cast(ctx, lhs, rhs):
    """Function::Execute's implicit casts (function.cc:175-176, cast.cc:228-243) behind CompareFunction /
    ArithmeticFunction::DispatchBest (scalar_compare.cc:75-98, scalar_arithmetic.cc:389-412)."""
    if not (isinstance(lhs, DeviceArray) and isinstance(rhs, DeviceArray)) or lhs.type_id == rhs.type_id:
        return lhs, rhs
    t = common_numeric(lhs.type_id, rhs.type_id)
    if t is None:
        raise NotImplementedError("no kernel matching input types (%d, %d)" % (lhs.type_id, rhs.type_id))
    return (lhs if lhs.type_id == t else cast(ctx, lhs, t)), (rhs if rhs.type_id == t else cast(ctx, rhs, t))


def compare(ctx, op, lhs, rhs):
    """equal / not_equal / greater / greater_equal / less / less_equal -- scalar_compare.cc:100-171."""
    lhs, rhs = _implicitly_cast(ctx, lhs, rhs)
    return _binary(ctx, ctx.lib.amdarrow_compare, op, lhs, rhs, BOOL)


def arith(ctx, op, lhs, rhs=None):
    """add / subtract / multiply / divide [+ _checked], power[_checked], negate[_checked] (unary: rhs unused) --
    scalar_arithmetic.cc:69-356."""
    if op >= _abi.NEGATE:
        dev, n = lhs.values.device, lhs.length
        out_v = alloc_bytes(_values_bytes(lhs.type_id, n), dev)
        out_b = alloc_bytes((n + 7) // 8, dev)
        out, c = COut(), lhs.c()
        check(ctx.lib, ctx.lib.amdarrow_arith(ctx.handle, op, C.byref(c), None, None, None, _ptr(out_v), _ptr(out_b), C.byref(out)))
        return DeviceArray(lhs.type_id, out_v, out_b, out.length, 0, out.null_count)
    lhs, rhs = _implicitly_cast(ctx, lhs, rhs)
    return _binary(ctx, ctx.lib.amdarrow_arith, op, lhs, rhs, None)


def _finish(ot, out_v, out_b, out):
    """DeviceArray of a scalar kernel whose validity buffer may be absent (has_validity == 0): the
    buffer (all ones then) stays reachable as .validity_buffer."""
    d = DeviceArray(ot, out_v, out_b if out.has_validity else None, out.length, 0, out.null_count)
    d.validity_buffer = out_b
    return d


def boolean(ctx, op, lhs, rhs=None):
    """invert / and / and_not / or / xor / and_kleene / and_not_kleene / or_kleene --
    scalar_boolean.cc:521-542.  lhs / rhs: a boolean DeviceArray, or a scalar: False, True, None (null)."""
    arr = lhs if isinstance(lhs, DeviceArray) else rhs
    n, dev = arr.length, arr.values.device
    keep = []

    def operand(x):
        if isinstance(x, DeviceArray):
            c = x.c()
            keep.append(c)
            return C.byref(c), _abi.SCALAR_NONE
        return None, (_abi.SCALAR_NULL if x is None else int(bool(x)))

    la, ls = operand(lhs)
    ra, rs = (None, _abi.SCALAR_TRUE) if op == _abi.INVERT else operand(rhs)
    out_v = alloc_bytes((n + 7) // 8, dev)
    out_b = alloc_bytes((n + 7) // 8, dev)
    out = COut()
    check(ctx.lib, ctx.lib.amdarrow_boolean(ctx.handle, op, la, ls, ra, rs, _ptr(out_v), _ptr(out_b), C.byref(out)))
    return _finish(BOOL, out_v, out_b, out)


def is_valid(ctx, a, is_null=False):
    """is_valid / is_null -- scalar_validity.cc:34-82 (result never null)."""
    out_v = alloc_bytes((a.length + 7) // 8, a.values.device)
    out = COut()
    c = a.c()
    check(ctx.lib, ctx.lib.amdarrow_validity_test(ctx.handle, 1 if is_null else 0, C.byref(c), _ptr(out_v), C.byref(out)))
    return DeviceArray(BOOL, out_v, None, out.length, 0, 0)


def is_null(ctx, a):
    return is_valid(ctx, a, is_null=True)


def is_nan(ctx, a):
    """is_nan -- scalar_validity.cc:84-89, 111-126."""
    n, dev = a.length, a.values.device
    out_v, out_b, out = alloc_bytes((n + 7) // 8, dev), alloc_bytes((n + 7) // 8, dev), COut()
    c = a.c()
    check(ctx.lib, ctx.lib.amdarrow_is_nan(ctx.handle, C.byref(c), _ptr(out_v), _ptr(out_b), C.byref(out)))
    return _finish(BOOL, out_v, out_b, out)


def fill_null(ctx, a, fill):
    """fill_null(values, scalar) -- scalar_fill_null.cc:44-147.  fill: a value of a's type or None (null scalar)."""
    n, dev, t = a.length, a.values.device, a.type_id
    out_v, out_b, out = alloc_bytes(_values_bytes(t, n), dev), alloc_bytes((n + 7) // 8, dev), COut()
    c = a.c()
    keep = None if fill is None else np.array([fill], dtype=np.uint8 if t == BOOL else NP_OF[t])
    fp = None if fill is None else keep.ctypes.data_as(C.c_void_p)
    check(ctx.lib, ctx.lib.amdarrow_fill_null(ctx.handle, C.byref(c), fp, _ptr(out_v), _ptr(out_b), C.byref(out)))
    return _finish(t, out_v, out_b, out)


def aggregate(ctx, a):
    """count / sum / min / max of one array in one pass -- aggregate_basic_internal.h:54-330.
    -> (valid_count, sum_bits, min_bits, max_bits): 64-bit patterns in the accumulator family."""
    out = _abi.CAgg()
    c = a.c()
    check(ctx.lib, ctx.lib.amdarrow_aggregate(ctx.handle, C.byref(c), C.byref(out)))
    return out.valid_count, out.sum, out.min, out.max


def make_groupings(ctx, ids, num_groups):
    """Grouper::MakeGroupings (hash_aggregate.cc:979-1018): ids = uint32 DeviceArray without nulls -> the two buffers
    (offsets int32[num_groups + 1], indices int32[n]) of the ListArray listing every group's rows in row order."""
    if ids.null_count != 0 and ids.validity is not None:
        raise AmdArrowError(4, "MakeGroupings with null ids")
    dev = ids.values.device
    n = ids.length
    off = torch.empty(num_groups + 1, dtype=torch.int32, device=dev)
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    check(ctx.lib, ctx.lib.amdarrow_make_groupings(ctx.handle, ids.values.data_ptr() + 4 * ids.offset, n, num_groups,
                                                   off.data_ptr(), idx.data_ptr()))
    return off, idx[:n]


def hash_aggregate(ctx, func, values, ids, num_groups, count_mode=0):
    """One HashAggregateKernel (hash_sum / hash_count / hash_mean) over (values, ids, num_groups).
    `ids`: torch.int32/uint32 tensor on the device."""
    dev = values.values.device
    out_v = alloc_bytes(num_groups * 8, dev)
    out_b = alloc_bytes((num_groups + 7) // 8, dev)
    out = COut()
    cv = values.c()
    check(ctx.lib, ctx.lib.amdarrow_hash_aggregate(ctx.handle, func, C.byref(cv), _ptr(ids), num_groups, count_mode,
                                                   _ptr(out_v), _ptr(out_b), C.byref(out)))
    t = values.type_id
    if func == _abi.HASH_COUNT:
        ot = _abi.INT64
    elif func == _abi.HASH_MEAN or t in (_abi.FLOAT, _abi.DOUBLE):
        ot = _abi.DOUBLE
    elif t in (_abi.INT8, _abi.INT16, _abi.INT32, _abi.INT64):
        ot = _abi.INT64
    else:
        ot = _abi.UINT64
    return DeviceArray(ot, out_v, out_b if out.has_validity else None, out.length, 0, out.null_count)


class HashMinMax:
    """hash_min_max's kernel state (GroupedMinMaxImpl, hash_aggregate.cc:627-768): resize / consume /
    finalize over any number of batches."""

    def __init__(self, ctx, value_type, device="cuda", null_handling=_abi.MINMAX_SKIP):
        self.ctx, self.t, self.dev, self.mode = ctx, value_type, device, null_handling
        self.num_groups, self.cap = 0, 0
        self.mins = self.maxes = self.has_values = self.has_nulls = None

    def _reserve(self, new_groups):  # MaybeReserve (:450-459) + the resize lambdas (:713-721)
        if new_groups > self.cap:
            cap = max(1024, 2 * new_groups)
            w = _values_bytes(self.t, 1)
            bufs = [alloc_bytes(cap * w, self.dev), alloc_bytes(cap * w, self.dev),
                    alloc_bytes((cap + 63) // 64 * 8, self.dev, zero=True), alloc_bytes((cap + 63) // 64 * 8, self.dev, zero=True)]
            for new, old in zip(bufs, (self.mins, self.maxes, self.has_values, self.has_nulls)):
                if old is not None:
                    new[:old.numel()] = old
            self.mins, self.maxes, self.has_values, self.has_nulls = bufs
            self.cap = cap
        if new_groups > self.num_groups:
            check(self.ctx.lib, self.ctx.lib.amdarrow_hash_minmax_init(self.ctx.handle, self.t, _ptr(self.mins), _ptr(self.maxes),
                                                                       self.num_groups, new_groups - self.num_groups))
            self.num_groups = new_groups

    def consume(self, values, ids, num_groups):
        self._reserve(num_groups)
        cv = values.c()
        check(self.ctx.lib, self.ctx.lib.amdarrow_hash_minmax_accumulate(self.ctx.handle, C.byref(cv), _ptr(ids), _ptr(self.mins),
                                                                         _ptr(self.maxes), _ptr(self.has_values),
                                                                         _ptr(self.has_nulls)))

    def finalize(self):
        """-> (mins, maxes): the two children of the reference's struct, one shared validity"""
        n = self.num_groups
        out_b = alloc_bytes((n + 63) // 64 * 8, self.dev, zero=True)
        out = COut()
        check(self.ctx.lib, self.ctx.lib.amdarrow_hash_minmax_finalize(self.ctx.handle, n, self.mode, _ptr(self.has_values),
                                                                       _ptr(self.has_nulls), _ptr(out_b), C.byref(out)))
        return (DeviceArray(self.t, self.mins, out_b, n, 0, out.null_count),
                DeviceArray(self.t, self.maxes, out_b, n, 0, out.null_count))


class Grouper:
    """internal::Grouper (api_aggregate.h:313-403): Consume(keys) -> uint32 ids in first-appearance order."""

    def __init__(self, ctx, key_type_ids):
        self.ctx, self.types = ctx, list(key_type_ids)
        h = C.c_void_p()
        arr = (C.c_int * len(self.types))(*self.types)
        check(ctx.lib, ctx.lib.amdarrow_grouper_create(ctx.handle, len(self.types), arr, C.byref(h)))
        self.handle = h

    def consume(self, keys):
        n = keys[0].length
        out = alloc_bytes(n * 4, keys[0].values.device)
        arr = (CArray * len(keys))(*[k.c() for k in keys])
        check(self.ctx.lib, self.ctx.lib.amdarrow_grouper_consume(self.handle, arr, _ptr(out)))
        return DeviceArray(_abi.UINT32, out, None, n, 0, 0)

    def num_groups(self):
        n = C.c_uint32(0)
        check(self.ctx.lib, self.ctx.lib.amdarrow_grouper_num_groups(self.handle, C.byref(n)))
        return n.value

    def uniques(self, k=0):
        ng = self.num_groups()
        t = self.types[k]
        out_v = alloc_bytes(_values_bytes(t, ng), self.ctx.device)
        out_b = alloc_bytes((ng + 7) // 8, self.ctx.device)
        out = COut()
        check(self.ctx.lib, self.ctx.lib.amdarrow_grouper_uniques(self.handle, k, _ptr(out_v), _ptr(out_b),
                                                                  C.byref(out)))
        return DeviceArray(t, out_v, out_b if out.has_validity else None, out.length, 0, out.null_count)

    def close(self):
        if self.handle:
            self.ctx.lib.amdarrow_grouper_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _group(ctx, arr):
    g = Grouper(ctx, [arr.type_id])
    try:
        ids = g.consume([arr])
        return ids, g.uniques(0), g.num_groups()
    finally:
        g.close()


def unique(ctx, arr):
    """CallFunction("unique") -- UniqueAction, vector_hash.cc:55-86: distinct values in order of first
    appearance, a null (once) included."""
    return _group(ctx, arr)[1]


def value_counts(ctx, arr):
    """CallFunction("value_counts") -- ValueCountsAction, vector_hash.cc:91-160: -> (values, counts int64),
    the two children of the reference's struct."""
    ids, uniq, ng = _group(ctx, arr)
    counts = hash_aggregate(ctx, _abi.HASH_COUNT, ids, ids.values, ng, _abi.COUNT_NON_NULL)  # every row counts, null keys too
    return uniq, counts


def dictionary_encode(ctx, arr, null_encoding=_abi.DICT_MASK):
    """CallFunction("dictionary_encode") -- DictEncodeAction, vector_hash.cc:165-225: -> (indices int32, dictionary)."""
    ids, uniq, ng = _group(ctx, arr)
    n = arr.length
    if null_encoding == _abi.DICT_ENCODE:  # the null is a dictionary entry; the ids are the indices
        return DeviceArray(_abi.INT32, ids.values, None, n, 0, 0), uniq
    out_i = alloc_bytes(4 * n, ctx.device)
    out_b = alloc_bytes((n + 63) // 64 * 8, ctx.device)
    out = COut()
    check(ctx.lib, ctx.lib.amdarrow_dictionary_indices(ctx.handle, _ptr(ids.values), n,
                                                       _ptr(uniq.validity) if uniq.validity is not None else None, ng,
                                                       _ptr(out_i), _ptr(out_b), C.byref(out)))
    indices = DeviceArray(_abi.INT32, out_i, out_b if out.has_validity else None, n, 0, out.null_count)
    if uniq.validity is not None:  # drop the null entry: Filter(uniques, is_valid(uniques))
        kept = filter(ctx, uniq, is_valid(ctx, uniq))
        uniq = DeviceArray(kept.type_id, kept.values, None, kept.length, 0, 0)  # a dictionary without nulls has no bitmap
    return indices, uniq


class GroupByResult:
    """StructArray{hash_sum, hash_count, key_0} of internal::GroupBy (hash_aggregate.cc:962-965),
    rows in first-appearance order, plus the first-row index of each group."""

    def __init__(self, num_groups, sums, counts, keys, key_validity, first_rows, value_type, key_type):
        self.num_groups, self.sums, self.counts = num_groups, sums, counts
        self.keys, self.key_validity, self.first_rows = keys, key_validity, first_rows
        self.value_type, self.key_type = value_type, key_type

    def sums_numpy(self):
        t = self.value_type
        acc = np.float64 if t in (_abi.FLOAT, _abi.DOUBLE) else (
            np.int64 if t in (_abi.INT8, _abi.INT16, _abi.INT32, _abi.INT64) else np.uint64)
        return self.sums[:self.num_groups * 8].cpu().numpy().view(acc).copy()

    def counts_numpy(self):
        return self.counts[:self.num_groups * 8].cpu().numpy().view(np.int64).copy()

    def keys_numpy(self):
        if self.key_type == BOOL:  # bit-packed
            raw = self.keys[:(self.num_groups + 7) // 8].cpu().numpy()
            return np.unpackbits(raw, bitorder="little")[:self.num_groups].astype(bool)
        w = bit_width(self.key_type) // 8
        return self.keys[:self.num_groups * w].cpu().numpy().view(NP_OF[self.key_type]).copy()

    def key_validity_numpy(self):
        raw = self.key_validity[:(self.num_groups + 7) // 8].cpu().numpy()
        return np.unpackbits(raw, bitorder="little")[:self.num_groups].astype(bool)

    def first_rows_numpy(self):
        return self.first_rows[:self.num_groups * 4].cpu().numpy().view(np.uint32).copy()


class GroupByBuffers:
    """Reusable output buffers for filter_groupby (so a timed loop allocates nothing)."""

    def __init__(self, device, group_capacity, key_type):
        self.cap = group_capacity
        self.sums = alloc_bytes(group_capacity * 8, device)
        self.counts = alloc_bytes(group_capacity * 8, device)
        self.keys = alloc_bytes(((group_capacity * bit_width(key_type) + 63) // 64) * 8, device)  # whole 8-byte words
        self.key_validity = alloc_bytes((group_capacity + 7) // 8, device)
        self.first_rows = alloc_bytes(group_capacity * 4, device)


def filter_groupby(ctx, values, keys, mask=None, group_capacity=None, buffers=None):
    """Filter(values, mask) + Filter(keys, mask) + GroupBy({values},{keys},{hash_sum, hash_count})
    fused into one pass over HBM (BASELINE.json north_star)."""
    cap = group_capacity if group_capacity is not None else (buffers.cap if buffers else values.length)
    b = buffers or GroupByBuffers(values.values.device, cap, keys.type_id)
    ng = C.c_int64(0)
    cv, ck = values.c(), keys.c()
    cm = mask.c() if mask is not None else None
    check(ctx.lib, ctx.lib.amdarrow_filter_groupby(ctx.handle, C.byref(cv), C.byref(ck),
                                                   C.byref(cm) if cm is not None else None, cap, _ptr(b.sums),
                                                   _ptr(b.counts), _ptr(b.keys), _ptr(b.key_validity),
                                                   _ptr(b.first_rows), C.byref(ng)))
    return GroupByResult(ng.value, b.sums, b.counts, b.keys, b.key_validity, b.first_rows, values.type_id,
                         keys.type_id)


def filter_groupby_multi(ctx, values, keys, mask=None, group_capacity=None):
    """The fused Filter -> GroupBy{hash_sum, hash_count} over several key columns (packed into one key word on the
    device).  -> (num_groups, sums DeviceArray, counts DeviceArray, [key DeviceArray per column], first_rows tensor).
    Raises AmdArrowError NotImplemented / CapacityError for shapes the packed path does not take."""
    n = keys[0].length
    cap = group_capacity if group_capacity is not None else n
    dev = keys[0].values.device
    sums, counts = alloc_bytes(cap * 8, dev), alloc_bytes(cap * 8, dev)
    first_rows = alloc_bytes(cap * 4, dev)
    kbufs = [alloc_bytes(((cap * bit_width(k.type_id) + 63) // 64) * 8, dev) for k in keys]
    vbufs = [alloc_bytes((cap + 63) // 64 * 8, dev) for _ in keys]
    karr = (CArray * len(keys))(*[k.c() for k in keys])
    kp = (C.c_void_p * len(keys))(*[b.data_ptr() for b in kbufs])
    vp = (C.c_void_p * len(keys))(*[b.data_ptr() for b in vbufs])
    nulls = (C.c_int64 * len(keys))()
    ng = C.c_int64(0)
    cv = values.c() if values is not None else None
    cm = mask.c() if mask is not None else None
    check(ctx.lib, ctx.lib.amdarrow_filter_groupby_multi(ctx.handle, C.byref(cv) if cv is not None else None, karr, len(keys),
                                                         C.byref(cm) if cm is not None else None, cap, _ptr(sums), _ptr(counts), kp, vp,
                                                         nulls, _ptr(first_rows), C.byref(ng)))
    G = ng.value
    acc = _abi.DOUBLE if values is not None and values.type_id in (_abi.FLOAT, _abi.DOUBLE) else (
        _abi.INT64 if values is None or values.type_id in (_abi.INT8, _abi.INT16, _abi.INT32, _abi.INT64) else _abi.UINT64)
    out_keys = [DeviceArray(k.type_id, kb, vb if nulls[i] else None, G, 0, nulls[i]) for i, (k, kb, vb) in enumerate(zip(keys, kbufs, vbufs))]
    return G, DeviceArray(acc, sums, None, G), DeviceArray(_abi.INT64, counts, None, G), out_keys, first_rows


_BOOL_OPS = {"invert": _abi.INVERT, "and": _abi.AND, "and_not": _abi.AND_NOT, "or": _abi.OR, "xor": _abi.XOR,
             "and_kleene": _abi.AND_KLEENE, "and_not_kleene": _abi.AND_NOT_KLEENE, "or_kleene": _abi.OR_KLEENE}


def call_function(ctx, name, args, options=None):
    """CallFunction(name, args, options, ctx) for the functions on the path (same names as the
    reference registry; unknown name -> KeyError like registry.cc:62-64)."""
    options = options or {}
    from .device import DeviceBinaryArray
    if name in ("filter", "array_filter"):
        if isinstance(args[0], DeviceBinaryArray):
            return binary_filter(ctx, args[0], args[1], options.get("null_selection_behavior", DROP))
        return filter(ctx, args[0], args[1], options.get("null_selection_behavior", DROP))
    if name in ("take", "array_take"):
        if isinstance(args[0], DeviceBinaryArray):
            return binary_take(ctx, args[0], args[1], options.get("boundscheck", True))
        return take(ctx, args[0], args[1], options.get("boundscheck", True))
    if name in ("sort_indices", "array_sort_indices"):
        if "sort_keys" in options:  # a batch / table: args[0] = {column name: chunks}, sort_keys = [(name, order), ...]
            keys = options["sort_keys"]
            return sort_indices_multi(ctx, [args[0][k] for k, _ in keys], [o for _, o in keys])
        return sort_indices(ctx, args[0], options.get("order", ASCENDING))
    simple = {"unique": unique, "value_counts": value_counts, "is_valid": is_valid, "is_nan": is_nan}
    if name in simple:
        return simple[name](ctx, args[0])
    if name == "is_null":
        return is_valid(ctx, args[0], is_null=True)
    if name == "dictionary_encode":
        return dictionary_encode(ctx, args[0], options.get("null_encoding_behavior", _abi.DICT_MASK))
    if name == "fill_null":
        return fill_null(ctx, args[0], args[1])
    if name in _BOOL_OPS:
        return boolean(ctx, _BOOL_OPS[name], args[0], args[1] if len(args) > 1 else None)
    if name in ("count", "sum", "mean", "min_max"):
        return aggregate(ctx, args[0])
    if name in _CMP:
        return compare(ctx, _CMP[name], args[0], args[1])
    if name in _ARITH:
        return arith(ctx, _ARITH[name], args[0], args[1] if len(args) > 1 else None)
    if name == "cast":  # options: {"to_type": type id, "safe": bool} (CastOptions)
        if not options or options.get("to_type") is None:
            raise AmdArrowError(4, "Cast requires that options be passed with the to_type populated")
        return cast(ctx, args[0], options["to_type"], options.get("safe", True))
    if name in _AGG:
        # function.cc:193-195: hash aggregates cannot be called directly
        raise AmdArrowError(10, "Direct execution of HASH_AGGREGATE functions")
    raise AmdArrowError(2, "No function registered with name: " + name)
