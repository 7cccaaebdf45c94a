"""array_take / array_filter for the value types that are not plain primitives (vector_selection.cc:2133-2176):
(large_)binary / utf8, fixed_size_binary / decimal, (large_)list / map, fixed_size_list, struct, dictionary, null --
the host side of the kernels in csrc/selection_nested.hip, composed exactly as the reference composes its Selection
implementations (child arrays are taken recursively with the generated child indices; struct fields with the same
selection).  Every byte moves in HBM through the C ABI; nothing here computes on the CPU."""
import ctypes as C

from . import _abi, compute
from ._abi import COut, CVarArray, AmdArrowError, check
from .device import DeviceArray, DeviceBinaryArray, alloc_bytes

(K_PRIM, K_BINARY, K_LARGE_BINARY, K_FSB, K_LIST, K_LARGE_LIST, K_MAP, K_FSL, K_STRUCT, K_DICT, K_NULL, K_DEC128, K_DEC256,
 K_UTF8, K_LARGE_UTF8) = range(15)
VAR32 = (K_BINARY, K_UTF8, K_LIST, K_MAP)
VAR64 = (K_LARGE_BINARY, K_LARGE_UTF8, K_LARGE_LIST)
BINARY_KINDS = (K_BINARY, K_UTF8, K_LARGE_BINARY, K_LARGE_UTF8)
LIST_KINDS = (K_LIST, K_MAP, K_LARGE_LIST)
FIXED_KINDS = (K_FSB, K_DEC128, K_DEC256)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class DeviceNested:
    """ArrayData of any of the types above with its buffers in HBM (torch uint8 tensors): `buf1` = values | offsets |
    fixed-width bytes | dictionary indices, `buf2` = binary data, `children` as the layout has them (DICT: [dictionary])."""

    def __init__(self, kind, length, offset=0, validity=None, null_count=0, buf1=None, buf2=None, children=(), type_id=0,
                 byte_width=0, list_size=0):
        self.kind, self.length, self.offset = kind, int(length), int(offset)
        self.validity, self.buf1, self.buf2, self.children = validity, buf1, buf2, list(children)
        self.type_id, self.list_size = type_id, list_size
        self.byte_width = 16 if kind == K_DEC128 else 32 if kind == K_DEC256 else byte_width
        self.null_count = self.length if kind == K_NULL else 0 if validity is None else int(null_count)

    def slice(self, offset, length):
        return DeviceNested(self.kind, length, self.offset + offset, self.validity, -1 if self.validity is not None else 0, self.buf1,
                            self.buf2, self.children, self.type_id, self.byte_width, self.list_size)

    def as_array(self):
        """PRIM values / DICT indices as a DeviceArray"""
        return DeviceArray(self.type_id, self.buf1, self.validity, self.length, self.offset, self.null_count)

    def as_binary(self):
        return DeviceBinaryArray(self.buf1, self.buf2, self.validity, self.length, self.offset, self.null_count)

    def var(self):
        s = CVarArray()
        s.offsets, s.offset_bits = self.buf1.data_ptr(), 32 if self.kind in VAR32 else 64
        s.data = None if self.buf2 is None else self.buf2.data_ptr()
        s.validity = None if self.validity is None else self.validity.data_ptr()
        s.length, s.offset, s.null_count = self.length, self.offset, self.null_count
        return s


def _take_rows(ctx, values, indices, boundscheck, dev):
    """Selection::VisitTake on the device: -> (source rows int64 tensor, validity tensor, COut)"""
    n = indices.length
    rows = alloc_bytes(8 * n, dev)
    bits = alloc_bytes((n + 63) // 64 * 8, dev)
    out, ci = COut(), indices.c()
    check(ctx.lib, ctx.lib.amdarrow_take_rows(ctx.handle, _ptr(values.validity), values.length, values.offset, values.null_count,
                                              C.byref(ci), int(boundscheck), _ptr(rows), _ptr(bits), C.byref(out)))
    return rows, bits, out


def take(ctx, values, indices, boundscheck=True):
    """CallFunction("take", {values, indices}) for a DeviceNested -- TakeExec<Impl>, vector_selection.cc:1330-1345."""
    k, n = values.kind, indices.length
    dev = indices.values.device
    if k == K_PRIM:  # PrimitiveTake :493-530
        r = compute.take(ctx, values.as_array(), indices, boundscheck)
        return DeviceNested(K_PRIM, r.length, 0, r.validity, r.null_count, r.values, type_id=values.type_id)
    if k in (K_BINARY, K_UTF8):  # VarBinaryImpl<BinaryType> :1493-1573
        r = compute.binary_take(ctx, values.as_binary(), indices, boundscheck)
        return DeviceNested(k, r.length, 0, r.validity, r.null_count, r.offsets, r.data)
    if k == K_DICT:  # DictionaryTake :1151-1160
        r = compute.take(ctx, values.as_array(), indices, boundscheck)
        return DeviceNested(K_DICT, r.length, 0, r.validity, r.null_count, r.values, children=values.children, type_id=values.type_id)
    if k == K_NULL:  # NullTake :1132-1140
        if boundscheck:
            _take_rows(ctx, values, indices, True, dev)
        return DeviceNested(K_NULL, n)
    rows, bits, out = _take_rows(ctx, values, indices, boundscheck, dev)
    if k in FIXED_KINDS:  # FSBImpl :1575-1606
        w = values.byte_width
        data = alloc_bytes(n * w, dev)
        check(ctx.lib, ctx.lib.amdarrow_gather_fixed_width(ctx.handle, _ptr(values.buf1), values.offset, w, _ptr(rows), n, _ptr(data)))
        return DeviceNested(k, n, 0, bits, out.null_count, data, byte_width=w)
    if k in (K_LARGE_BINARY, K_LARGE_UTF8) or k in LIST_KINDS:
        ow = 4 if k in VAR32 else 8
        offs = alloc_bytes(ow * (n + 1), dev, zero=True)
        cv, total = values.var(), C.c_int64(0)
        check(ctx.lib, ctx.lib.amdarrow_var_take_offsets(ctx.handle, C.byref(cv), _ptr(rows), n, _ptr(offs), C.byref(total)))
        if k in BINARY_KINDS:  # VarBinaryImpl<LargeBinaryType>
            data = alloc_bytes(total.value, dev)
            check(ctx.lib, ctx.lib.amdarrow_var_take_data(ctx.handle, C.byref(cv), _ptr(rows), n, _ptr(offs), _ptr(data)))
            return DeviceNested(k, n, 0, bits, out.null_count, offs, data)
        # ListImpl :1608-1680: the child is Take(values.values(), child_indices, NoBoundsCheck)
        cidx = alloc_bytes(ow * total.value, dev)
        check(ctx.lib, ctx.lib.amdarrow_list_take_child_indices(ctx.handle, C.byref(cv), _ptr(rows), n, _ptr(offs), _ptr(cidx)))
        child = take(ctx, values.children[0], DeviceArray(_abi.INT32 if ow == 4 else _abi.INT64, cidx, None, total.value), False)
        return DeviceNested(k, n, 0, bits, out.null_count, offs, children=[child])
    if k == K_FSL:  # FSLImpl :1682-1724
        ls = values.list_size
        cidx = alloc_bytes(8 * n * ls, dev)
        cbits = alloc_bytes((n * ls + 63) // 64 * 8, dev)
        check(ctx.lib, ctx.lib.amdarrow_fixed_size_list_child_indices(ctx.handle, _ptr(rows), n, ls, values.offset, _ptr(cidx), _ptr(cbits)))
        child_idx = DeviceArray(_abi.INT64, cidx, cbits if out.null_count else None, n * ls, 0, out.null_count * ls)
        return DeviceNested(K_FSL, n, 0, bits, out.null_count, children=[take(ctx, values.children[0], child_idx, False)], list_size=ls)
    if k == K_STRUCT:  # StructImpl :1736-1763: the validity is the kernel's whole output; the fields take the same selection
        kids = [take(ctx, c.slice(values.offset, values.length), indices, False) for c in values.children]
        return DeviceNested(K_STRUCT, n, 0, bits, out.null_count, children=kids)
    raise AmdArrowError(10, "take: values of this type")


def filter(ctx, values, mask, null_selection=_abi.DROP):  # noqa: A001 (reference name)
    """CallFunction("filter", {values, mask}) for a DeviceNested -- the array_filter kernels, :2133-2148."""
    k = values.kind
    if k == K_DEC256:  # array_filter has no decimal256 kernel (Type::DECIMAL is DECIMAL128, :2139)
        raise AmdArrowError(10, "Function array_filter has no kernel matching input types (array[decimal256(40, 3)], array[bool])")
    if k == K_PRIM:
        r = compute.filter(ctx, values.as_array(), mask, null_selection)
        return DeviceNested(K_PRIM, r.length, 0, r.validity, r.null_count, r.values, type_id=values.type_id)
    if k in (K_BINARY, K_UTF8):
        r = compute.binary_filter(ctx, values.as_binary(), mask, null_selection)
        return DeviceNested(k, r.length, 0, r.validity, r.null_count, r.offsets, r.data)
    if k == K_DICT:  # DictionaryFilter :1162-1172
        r = compute.filter(ctx, values.as_array(), mask, null_selection)
        return DeviceNested(K_DICT, r.length, 0, r.validity, r.null_count, r.values, children=values.children, type_id=values.type_id)
    if k == K_NULL:  # NullFilter :1142-1147
        return DeviceNested(K_NULL, compute.filter_output_size(ctx, mask, null_selection))
    # FilterExec<Impl> visits the rows GetTakeIndices lists, in that order (StructFilter :1765-1781 says so itself)
    r = take(ctx, values, compute.take_indices(ctx, mask, null_selection), False)
    if k in (K_LARGE_BINARY, K_LARGE_UTF8):  # BinaryFilter's conventions, :1090-1112
        vnc = values.null_count if values.validity is not None else 0
        fnc = mask.null_count if mask.validity is not None else 0
        if vnc == 0 and fnc == 0:
            r.validity, r.null_count = None, 0
        else:
            r.null_count = 0 if (vnc == 0 and null_selection == _abi.DROP) else -1
    return r


class BinaryGrouper:
    """Grouper over one (large_)binary / utf8 key column: VarLengthKeyEncoder, hash_aggregate.cc:237-311."""

    def __init__(self, ctx, kind):
        self.ctx, self.kind = ctx, kind
        h = C.c_void_p()
        check(ctx.lib, ctx.lib.amdarrow_binary_grouper_create(ctx.handle, 32 if kind in VAR32 else 64, C.byref(h)))
        self.handle = h

    def consume(self, keys):
        out = alloc_bytes(keys.length * 4, self.ctx.device)
        cv = keys.var()
        check(self.ctx.lib, self.ctx.lib.amdarrow_binary_grouper_consume(self.handle, C.byref(cv), _ptr(out)))
        return DeviceArray(_abi.UINT32, out, None, keys.length, 0, 0)

    def num_groups(self):
        n = C.c_uint32(0)
        check(self.ctx.lib, self.ctx.lib.amdarrow_binary_grouper_num_groups(self.handle, C.byref(n)))
        return n.value

    def uniques(self):
        rows, nbytes = C.c_int64(0), C.c_int64(0)
        check(self.ctx.lib, self.ctx.lib.amdarrow_binary_grouper_uniques_size(self.handle, C.byref(rows), C.byref(nbytes)))
        ow = 4 if self.kind in VAR32 else 8
        offs = alloc_bytes(ow * (rows.value + 1), self.ctx.device, zero=True)
        data = alloc_bytes(nbytes.value, self.ctx.device)
        bits = alloc_bytes((rows.value + 63) // 64 * 8, self.ctx.device)
        out = COut()
        check(self.ctx.lib, self.ctx.lib.amdarrow_binary_grouper_uniques(self.handle, _ptr(offs), _ptr(data), _ptr(bits), C.byref(out)))
        return DeviceNested(self.kind, out.length, 0, bits if out.has_validity else None, out.null_count, offs, data)

    def close(self):
        if self.handle:
            self.ctx.lib.amdarrow_binary_grouper_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Grouper:
    """internal::Grouper over key columns of any of the types GrouperImpl::Make accepts (hash_aggregate.cc:313-356):
    fixed-width and boolean columns as they are, dictionary columns by their indices (DictionaryKeyEncoder :198-235: one
    dictionary per key, "Unifying differing dictionaries" is NotImplemented there too), variable-width columns through a
    BinaryGrouper each, whose dense ids stand in for the column in the fixed-width Grouper over all columns -- equal rows
    <=> equal tuples, and that grouper numbers the tuples in first-appearance order."""

    def __init__(self, ctx, key_kinds_types):
        """key_kinds_types: per key column (kind, type_id) -- type_id for PRIM columns and dictionary indices"""
        self.ctx, self.cols = ctx, list(key_kinds_types)
        self.bin = [BinaryGrouper(ctx, k) if k in BINARY_KINDS else None for k, _ in self.cols]
        self.dicts = [None] * len(self.cols)
        for k, _ in self.cols:
            if k not in BINARY_KINDS + (K_PRIM, K_DICT):
                raise AmdArrowError(10, "Keys of type " + str(k))  # GrouperImpl::Make :353
        self.inner = compute.Grouper(ctx, [_abi.UINT32 if k in BINARY_KINDS else t for k, t in self.cols])

    def _same_dictionary(self, a, b):
        if a.kind != b.kind or a.length != b.length or a.type_id != b.type_id:
            return False
        if a.buf1 is b.buf1 and a.buf2 is b.buf2 and a.validity is b.validity and a.offset == b.offset:
            return True
        # Array::Equals on two small arrays: compare on the device through equal() + the validity bits
        import torch

        def logical(d):
            if d.kind in BINARY_KINDS:
                return d.as_binary().to_items()
            arr = d.as_array()
            v = arr.validity_numpy()
            return [x if ok else None for x, ok in zip(arr.values_numpy().tolist(), v)]
        del torch
        return logical(a) == logical(b)

    def consume(self, keys):
        fixed = []
        for c, key in enumerate(keys):
            kind = self.cols[c][0]
            if kind in BINARY_KINDS:
                fixed.append(self.bin[c].consume(key))
            elif kind == K_DICT:
                if self.dicts[c] is None:
                    self.dicts[c] = key.children[0]
                elif not self._same_dictionary(self.dicts[c], key.children[0]):
                    raise AmdArrowError(10, "Unifying differing dictionaries")
                fixed.append(key.as_array())
            else:
                fixed.append(key.as_array())
        return self.inner.consume(fixed)

    def num_groups(self):
        return self.inner.num_groups()

    def uniques(self, c):
        kind, t = self.cols[c]
        u = self.inner.uniques(c)
        if kind in BINARY_KINDS:  # the unique tuples name the column's keys by id: gather them from the column's store
            r = take(self.ctx, self.bin[c].uniques(), u, False) if u.length else self.bin[c].uniques()
            if r.null_count == 0:  # DecodeNulls :64-94: a validity bitmap only when a key is null
                r.validity = None
            return r
        if kind == K_DICT:
            d = self.dicts[c]
            return DeviceNested(K_DICT, u.length, 0, u.validity, u.null_count, u.values, children=[d] if d is not None else [], type_id=t)
        return DeviceNested(K_PRIM, u.length, 0, u.validity, u.null_count, u.values, type_id=t)

    def close(self):
        for b in self.bin:
            if b is not None:
                b.close()
        self.inner.close()
