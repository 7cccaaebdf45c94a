"""Arrow arrays of the non-primitive value types array_take / array_filter accept (vector_selection.cc:2133-2176) as
numpy trees, with three implementations of `select(values, selection, is_filter, option)` behind one interface:

  RefNested     the REAL reference through oracle/_ref/libref_driver.so (build container only; fixture generator)
  OracleNested  a numpy restatement of the reference's Selection kernels, each step citing the lines it follows
  (the HIP path is arrow-1_amd/nested.py, reached through tests/gpu_api.py)

TEST INFRASTRUCTURE: nothing under arrow-1_amd/ imports this."""
import ctypes as C
import os

import numpy as np

import orc
from orc import Arr, BOOL, INT32, INT64, NP_OF, pack, unpack

(K_PRIM, K_BINARY, K_LARGE_BINARY, K_FSB, K_LIST, K_LARGE_LIST, K_MAP, K_FSL, K_STRUCT, K_DICT, K_NULL, K_DEC128, K_DEC256,
 K_UTF8, K_LARGE_UTF8) = range(15)
KIND_NAMES = ["prim", "binary", "large_binary", "fsb", "list", "large_list", "map", "fsl", "struct", "dict", "null", "decimal128",
              "decimal256", "utf8", "large_utf8"]
VAR32 = (K_BINARY, K_UTF8, K_LIST, K_MAP)
VAR64 = (K_LARGE_BINARY, K_LARGE_UTF8, K_LARGE_LIST)
BINARY_KINDS = (K_BINARY, K_UTF8, K_LARGE_BINARY, K_LARGE_UTF8)
LIST_KINDS = (K_LIST, K_MAP, K_LARGE_LIST)
FIXED_KINDS = (K_FSB, K_DEC128, K_DEC256)


class CNode(C.Structure):
    _fields_ = [("kind", C.c_int32), ("type_id", C.c_int32), ("byte_width", C.c_int32), ("list_size", C.c_int32),
                ("buf1", C.c_void_p), ("buf2", C.c_void_p), ("validity", C.c_void_p), ("length", C.c_int64), ("offset", C.c_int64),
                ("null_count", C.c_int64), ("buf1_bytes", C.c_int64), ("buf2_bytes", C.c_int64), ("nchildren", C.c_int32),
                ("first_child", C.c_int32)]


class N:
    """One array: `buf1` = values (PRIM: typed numpy, bool array for BOOL) | offsets (int32 / int64) | fixed-width bytes
    (uint8, rows x byte_width) | dictionary indices; `buf2` = binary data bytes; `valid` = bool array over the PHYSICAL
    rows (offset + length) or None; children as the layout has them (DICT: [dictionary])."""

    def __init__(self, kind, length, offset=0, valid=None, null_count=None, buf1=None, buf2=None, children=(), type_id=0,
                 byte_width=0, list_size=0):
        self.kind, self.length, self.offset = kind, int(length), int(offset)
        self.valid = None if valid is None else np.asarray(valid, bool)
        self.buf1, self.buf2, self.children = buf1, buf2, list(children)
        self.type_id, self.byte_width, self.list_size = type_id, byte_width, list_size
        if kind == K_DEC128:
            self.byte_width = 16
        if kind == K_DEC256:
            self.byte_width = 32
        if kind == K_NULL:
            self.null_count = self.length
        elif self.valid is None:
            self.null_count = 0
        else:
            self.null_count = int((~self.valid[self.offset:self.offset + self.length]).sum()) if null_count is None else null_count

    def validity(self):
        if self.kind == K_NULL:
            return np.zeros(self.length, bool)
        return np.ones(self.length, bool) if self.valid is None else self.valid[self.offset:self.offset + self.length]

    # ---- logical value (python objects) --------------------------------------------------------------------------
    def to_py(self):
        v = self.validity()
        o = self.offset
        out = []
        kids = [c.to_py() for c in self.children] if self.kind in LIST_KINDS + (K_FSL, K_STRUCT, K_DICT) else []
        for i in range(self.length):
            if not v[i]:
                out.append(None)
            elif self.kind == K_PRIM:
                x = self.buf1[o + i]
                out.append(x.item() if not (isinstance(x.item(), float) and x != x) else "nan")
            elif self.kind in BINARY_KINDS:
                out.append(bytes(self.buf2[self.buf1[o + i]:self.buf1[o + i + 1]]))
            elif self.kind in FIXED_KINDS:
                out.append(bytes(self.buf1[(o + i) * self.byte_width:(o + i + 1) * self.byte_width]))
            elif self.kind in LIST_KINDS:
                out.append(kids[0][self.buf1[o + i]:self.buf1[o + i + 1]])
            elif self.kind == K_FSL:
                out.append(kids[0][(o + i) * self.list_size:(o + i + 1) * self.list_size])
            elif self.kind == K_STRUCT:
                out.append(tuple(k[o + i] for k in kids))
            elif self.kind == K_DICT:
                out.append(("dict", kids[0][int(self.buf1[o + i])]))
        return out

    # ---- physical comparison: what the reference's tests call AssertArraysEqual + the buffer conventions --------
    def same(self, other, path="root"):
        """-> (ok, message): every buffer the two arrays expose over their logical range, bit for bit."""
        if (self.kind, self.length, self.null_count, self.type_id, self.byte_width, self.list_size, len(self.children)) != \
                (other.kind, other.length, other.null_count, other.type_id, other.byte_width, other.list_size, len(other.children)):
            return False, "%s: shape %s vs %s" % (path, self.describe(), other.describe())
        if (self.valid is None) != (other.valid is None) and self.kind != K_NULL and self.length > 0:
            # (an empty bitmap is a zero-size buffer on one side and no buffer on the other: nothing to tell apart)
            return False, "%s: validity buffer present on one side only" % path
        if not np.array_equal(self.validity(), other.validity()):
            return False, "%s: validity differs" % path
        a0, b0 = self.offset, other.offset
        n = self.length
        if self.kind == K_PRIM or self.kind == K_DICT:
            x, y = self.buf1[a0:a0 + n], other.buf1[b0:b0 + n]
            if self.kind == K_DICT:  # index values under nulls are unspecified bytes of the primitive take: compared where valid
                m = self.validity()
                x, y = x[m], y[m]
            if not np.array_equal(np.asarray(x).view(np.uint8) if x.dtype != bool else x, np.asarray(y).view(np.uint8) if y.dtype != bool else y):
                return False, "%s: values differ" % path
        elif self.kind in FIXED_KINDS:
            w = self.byte_width
            if not np.array_equal(self.buf1[a0 * w:(a0 + n) * w], other.buf1[b0 * w:(b0 + n) * w]):
                return False, "%s: fixed-width bytes differ" % path
        elif self.kind in BINARY_KINDS + LIST_KINDS:
            x = self.buf1[a0:a0 + n + 1].astype(np.int64) - int(self.buf1[a0]) if len(self.buf1) else np.zeros(0, np.int64)
            y = other.buf1[b0:b0 + n + 1].astype(np.int64) - int(other.buf1[b0]) if len(other.buf1) else np.zeros(0, np.int64)
            if not np.array_equal(x, y):
                return False, "%s: offsets differ" % path
            if self.kind in BINARY_KINDS and n:
                if not np.array_equal(self.buf2[self.buf1[a0]:self.buf1[a0 + n]], other.buf2[other.buf1[b0]:other.buf1[b0 + n]]):
                    return False, "%s: data bytes differ" % path
        if self.kind in LIST_KINDS and n:
            ca, cb = self.children[0].slice(int(self.buf1[a0]), int(self.buf1[a0 + n] - self.buf1[a0])), \
                other.children[0].slice(int(other.buf1[b0]), int(other.buf1[b0 + n] - other.buf1[b0]))
            return ca.same(cb, path + ".values")
        if self.kind == K_FSL:
            return self.children[0].slice(a0 * self.list_size, n * self.list_size).same(
                other.children[0].slice(b0 * other.list_size, n * other.list_size), path + ".values")
        if self.kind == K_STRUCT:
            for k, (ca, cb) in enumerate(zip(self.children, other.children)):
                ok, msg = ca.slice(a0, n).same(cb.slice(b0, n), "%s.f%d" % (path, k))
                if not ok:
                    return ok, msg
        if self.kind == K_DICT:
            return self.children[0].same(other.children[0], path + ".dictionary")
        return True, ""

    def slice(self, start, length):
        """a view of rows [start, start + length) of the LOGICAL array (null_count recomputed)"""
        return N(self.kind, length, self.offset + start, self.valid, None, self.buf1, self.buf2, self.children, self.type_id,
                 self.byte_width, self.list_size)

    def describe(self):
        return "%s(len=%d nulls=%d t=%d w=%d ls=%d kids=%d)" % (KIND_NAMES[self.kind], self.length, self.null_count, self.type_id,
                                                                self.byte_width, self.list_size, len(self.children))

    # ---- to the C node list ------------------------------------------------------------------------------------
    def flatten(self):
        """-> (ctypes array of CNode, keepalive list)"""
        nodes, keep = [None], []

        def fill(at, a):
            c = CNode()
            c.kind, c.type_id, c.byte_width, c.list_size = a.kind, a.type_id, a.byte_width, a.list_size
            c.length, c.offset, c.null_count = a.length, a.offset, a.null_count
            if a.valid is not None:
                vb = np.concatenate([pack(a.valid), np.zeros(16, np.uint8)])
                keep.append(vb)
                c.validity = vb.ctypes.data
            if a.buf1 is not None:
                b = pack(a.buf1) if a.buf1.dtype == bool else np.ascontiguousarray(a.buf1)
                b = np.concatenate([b.view(np.uint8).ravel(), np.zeros(16, np.uint8)])
                keep.append(b)
                c.buf1 = b.ctypes.data
            if a.buf2 is not None:
                b = np.concatenate([np.ascontiguousarray(a.buf2, np.uint8), np.zeros(16, np.uint8)])
                keep.append(b)
                c.buf2 = b.ctypes.data
            c.nchildren, c.first_child = len(a.children), len(nodes)
            nodes[at] = c
            first = len(nodes)
            nodes.extend([None] * len(a.children))
            for k, ch in enumerate(a.children):
                fill(first + k, ch)
        fill(0, self)
        return (CNode * len(nodes))(*nodes), keep


def _read(ptr, nbytes):
    return np.frombuffer(C.string_at(ptr, nbytes), np.uint8).copy() if ptr and nbytes > 0 else np.zeros(0, np.uint8)


class RefNested:
    """take / filter / Grouper of the real reference over node trees."""

    def __init__(self):
        path = os.path.join(orc.ROOT, "oracle", "_ref", "libref_driver.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.lib.ref_last_error.restype = C.c_char_p
        self.lib.ref_nested_grouper_new.restype = C.c_void_p
        self.lib.ref_nested_grouper_num_groups.restype = C.c_uint32
        self.lib.ref_nested_grouper_num_groups.argtypes = [C.c_void_p]
        self.lib.ref_nested_grouper_free.argtypes = [C.c_void_p]

    def _check(self, rc):
        if rc != 0:
            raise orc.StatusError(rc, self.lib.ref_last_error().decode())

    def _held(self):
        cnt = self.lib.ref_nested_result_count()
        raw = []
        for i in range(cnt):
            c = CNode()
            assert self.lib.ref_nested_result_node(i, C.byref(c)) == 0
            raw.append(c)

        def build(i):
            c = raw[i]
            end = c.offset + c.length
            kids = [build(c.first_child + k) for k in range(c.nchildren)]
            valid = None
            if c.validity:
                valid = unpack(_read(c.validity, (end + 7) // 8), end)
            buf1 = buf2 = None
            if c.kind == K_PRIM or c.kind == K_DICT:
                b = _read(c.buf1, c.buf1_bytes)
                buf1 = unpack(b, end) if c.type_id == BOOL else b[:end * np.dtype(NP_OF[c.type_id]).itemsize].view(NP_OF[c.type_id])
            elif c.kind in VAR32:
                buf1 = _read(c.buf1, c.buf1_bytes)[:(end + 1) * 4].view(np.int32) if c.buf1 else np.zeros(1, np.int32)
                if len(buf1) == 0:
                    buf1 = np.zeros(1, np.int32)
            elif c.kind in VAR64:
                buf1 = _read(c.buf1, c.buf1_bytes)[:(end + 1) * 8].view(np.int64) if c.buf1 else np.zeros(1, np.int64)
                if len(buf1) == 0:
                    buf1 = np.zeros(1, np.int64)
            elif c.kind in FIXED_KINDS:
                buf1 = _read(c.buf1, c.buf1_bytes)
            if c.kind in BINARY_KINDS:
                buf2 = _read(c.buf2, c.buf2_bytes)
            return N(c.kind, c.length, c.offset, valid, c.null_count, buf1, buf2, kids, c.type_id, c.byte_width, c.list_size)
        return build(0)

    def select(self, values, sel, is_filter, option):
        nodes, keep = values.flatten()
        cs = sel.c()
        self._check(self.lib.ref_nested_select(nodes, len(nodes), C.byref(cs), int(is_filter), int(option)))
        return self._held()

    def group(self, batches):
        """batches: list of lists of N (key columns) -> (ids per batch, num_groups, uniques per column as N)"""
        nkeys = len(batches[0])

        def pack_cols(cols):
            allnodes, keep, roots = [], [], []
            for col in cols:
                nodes, k = col.flatten()
                base = len(allnodes)
                roots.append(base)
                for nd in nodes:
                    c = CNode()
                    C.memmove(C.byref(c), C.byref(nd), C.sizeof(CNode))
                    c.first_child += base
                    allnodes.append(c)
                keep.append(k)
            return (CNode * len(allnodes))(*allnodes), (C.c_int * nkeys)(*roots), keep
        nodes, roots, keep = pack_cols(batches[0])
        g = self.lib.ref_nested_grouper_new(nodes, len(nodes), roots, nkeys)
        if not g:
            raise orc.StatusError(orc.NOT_IMPLEMENTED, self.lib.ref_last_error().decode())
        try:
            ids, alive = [], [keep]  # (the reference keeps the first batch's dictionary: its buffers must outlive the loop)
            for cols in batches:
                nodes, roots, keep = pack_cols(cols)
                alive.append((nodes, keep))
                out = np.zeros(cols[0].length + 1, np.uint32)
                self._check(self.lib.ref_nested_grouper_consume(C.c_void_p(g), nodes, len(nodes), roots, out.ctypes.data_as(C.c_void_p)))
                ids.append(out[:cols[0].length].copy())
            ng = self.lib.ref_nested_grouper_num_groups(C.c_void_p(g))
            uniq = []
            for k in range(nkeys):
                self._check(self.lib.ref_nested_grouper_uniques(C.c_void_p(g), k))
                uniq.append(self._held())
            return ids, ng, uniq
        finally:
            self.lib.ref_nested_grouper_free(C.c_void_p(g))


# ---- the numpy restatement ------------------------------------------------------------------------------------------
def take_rows(values, idx, boundscheck):
    """Selection::VisitTake, vector_selection.cc:1214-1262 (+ CheckIndexBounds, util/int_util.cc:534-561, run by TakeExec
    :1330-1345 before anything else): -> (src rows, -1 = null; validity bool array).  An output row is valid iff its index is
    valid and the value it names is valid; out-of-range indices under a null are never looked at."""
    iv = idx.validity()
    raw = idx.values()
    if boundscheck:
        for k in range(idx.length):
            if iv[k] and (int(raw[k]) < 0 or int(raw[k]) >= values.length):
                raise orc.StatusError(orc.INDEX_ERROR, "Index %d out of bounds" % int(raw[k]))
    src = np.full(idx.length, -1, np.int64)
    vv = values.validity()
    for k in range(idx.length):
        if iv[k]:
            j = int(raw[k])
            if 0 <= j < values.length and vv[j]:
                src[k] = j
    return src, src >= 0


def take_indices_of(filt, null_sel):
    """GetTakeIndices, vector_selection.cc:92-237, as (index Arr): DROP keeps the rows whose filter slot is valid and
    true; EMIT_NULL also emits a NULL index for every null filter slot."""
    fv, fb = filt.validity(), filt.values()
    idx, valid = [], []
    for i in range(filt.length):
        if fv[i] and fb[i]:
            idx.append(i)
            valid.append(True)
        elif not fv[i] and null_sel == orc.EMIT_NULL:
            idx.append(0)
            valid.append(False)
    t = orc.UINT16 if filt.length <= 65535 else orc.UINT32
    return Arr(t, np.array(idx, NP_OF[t]), None if all(valid) else np.array(valid, bool))


_C_ORACLE = None


def _c_oracle():
    global _C_ORACLE
    if _C_ORACLE is None:
        _C_ORACLE = orc.Api("orc")
    return _C_ORACLE


class OracleNested:
    def select(self, values, sel, is_filter, option):
        if is_filter:
            k = values.kind
            if k == K_DEC256:  # array_filter registers Type::DECIMAL (= DECIMAL128) only, :2139
                raise orc.StatusError(orc.NOT_IMPLEMENTED, "Function array_filter has no kernel matching input types (array[decimal256(40, 3)], array[bool])")
            if k == K_NULL:  # NullFilter :1142-1147
                return N(K_NULL, take_indices_of(sel, option).length)
            if k == K_PRIM:  # PrimitiveFilter :786-835, restated in C (oracle/oracle.c)
                r = _c_oracle().filter(Arr(values.type_id, values.buf1, values.valid, values.offset, values.length, values.null_count), sel, option)
                return N(K_PRIM, r.length, 0, r.validity() if r.has_validity else None, r.null_count, r.values, type_id=values.type_id)
            if k == K_DICT:  # DictionaryFilter :1162-1172: Filter(indices), the dictionary untouched
                ind = N(K_PRIM, values.length, values.offset, values.valid, values.null_count, values.buf1, type_id=values.type_id)
                t = self.select(ind, sel, True, option)
                return N(K_DICT, t.length, 0, t.valid, t.null_count, t.buf1, children=values.children, type_id=values.type_id)
            # FilterExec<Impl> visits exactly the rows GetTakeIndices lists, in order (VisitFilter :1264-1328; StructFilter
            # :1765-1781 does it literally)
            r = self.take(values, take_indices_of(sel, option), False)
            if k in BINARY_KINDS:  # BinaryFilter's conventions for the validity buffer and the null count, :1090-1112
                vnc = values.null_count if values.valid is not None else 0
                fnc = sel.null_count if sel.valid is not None else 0
                if vnc == 0 and fnc == 0:
                    r.valid, r.null_count = None, 0
                else:
                    r.null_count = 0 if (vnc == 0 and option == orc.DROP) else -1
            return r
        return self.take(values, sel, bool(option))

    def take(self, values, idx, boundscheck):
        k = values.kind
        n = idx.length
        if k == K_NULL:  # NullTake :1132-1140
            if boundscheck:
                take_rows(values, idx, True)
            return N(K_NULL, n)
        if k == K_DICT:  # DictionaryTake :1151-1160: Take(indices), the dictionary untouched
            ind = N(K_PRIM, values.length, values.offset, values.valid, values.null_count, values.buf1, type_id=values.type_id)
            t = self.take(ind, idx, boundscheck)
            return N(K_DICT, n, 0, t.valid, t.null_count, t.buf1, children=values.children, type_id=values.type_id)
        src, ok = take_rows(values, idx, boundscheck)
        valid = ok.copy()  # FinishCommon :1205-1210: validity always there, null count exact
        o = values.offset
        if k == K_PRIM:  # PrimitiveTakeImpl :273-369 (null slots hold zero)
            dt = bool if values.type_id == BOOL else NP_OF[values.type_id]
            out = np.zeros(n, dt)
            out[ok] = values.buf1[o + src[ok]]
            return N(K_PRIM, n, 0, valid, None, out, type_id=values.type_id)
        if k in FIXED_KINDS:  # FSBImpl :1575-1606
            w = values.byte_width
            out = np.zeros(n * w, np.uint8)
            rows = values.buf1.reshape(-1, w) if w else np.zeros((len(src), 0), np.uint8)
            if w:
                out.reshape(-1, w)[ok] = rows[o + src[ok]]
            return N(k, n, 0, valid, None, out, byte_width=w)
        if k in BINARY_KINDS + LIST_KINDS:  # VarBinaryImpl :1493-1573 / ListImpl :1608-1680: a null slot is empty
            odt = np.int32 if k in VAR32 else np.int64
            lens = np.zeros(n, np.int64)
            lens[ok] = values.buf1[o + src[ok] + 1].astype(np.int64) - values.buf1[o + src[ok]]
            total = int(lens.sum())
            if k in (K_BINARY, K_UTF8) and total > 2**31 - 2:
                raise orc.StatusError(orc.INVALID, "Take operation overflowed binary array capacity")
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(odt)
            starts = np.zeros(n, np.int64)
            starts[ok] = values.buf1[o + src[ok]]
            run = np.concatenate([np.arange(s, s + l) for s, l in zip(starts, lens)]) if total else np.zeros(0, np.int64)
            if k in BINARY_KINDS:
                return N(k, n, 0, valid, None, offs, values.buf2[run] if total else np.zeros(0, np.uint8))
            child_idx = Arr(INT32 if k in VAR32 else INT64, run.astype(np.int32 if k in VAR32 else np.int64))
            return N(k, n, 0, valid, None, offs, children=[self.take(values.children[0], child_idx, False)])
        if k == K_FSL:  # FSLImpl :1682-1724 (with the array's offset applied, see include/amdarrow.h)
            ls = values.list_size
            ci = np.zeros(n * ls, np.int64)
            cv = np.repeat(ok, ls)
            base = np.repeat((o + np.where(ok, src, 0)) * ls, ls) + np.tile(np.arange(ls), n)
            ci[cv] = base[cv]
            child_idx = Arr(INT64, ci, None if cv.all() else cv)
            return N(K_FSL, n, 0, valid, None, children=[self.take(values.children[0], child_idx, False)], list_size=ls)
        if k == K_STRUCT:  # StructImpl :1736-1763: every field taken with the SAME selection (its nulls included)
            kids = [self.take(c.slice(o, values.length), idx, False) for c in values.children]
            return N(K_STRUCT, n, 0, valid, None, children=kids)
        raise NotImplementedError(KIND_NAMES[k])


def oracle_group(batches):
    """GrouperImpl::Consume / GetUniques, hash_aggregate.cc:358-437, over key columns given as trees: a row's key is the
    concatenation of its columns' encodings -- null flag + bytes (fixed-width :142-196, boolean :96-140), null flag +
    length + bytes (variable-width :237-311), the dictionary INDEX (:198-235) -- ids in first-appearance order.
    -> (ids per batch, num_groups, uniques per column as N)"""
    seen, firsts = {}, []
    ids_out = []
    dicts = [None] * len(batches[0])
    for cols in batches:
        n = cols[0].length
        for c, col in enumerate(cols):
            if col.kind == K_DICT:
                d = col.children[0].to_py()
                if dicts[c] is None:
                    dicts[c] = (d, col.children[0])
                elif dicts[c][0] != d:
                    raise orc.StatusError(orc.NOT_IMPLEMENTED, "Unifying differing dictionaries")
        keys = []
        for col in cols:
            v = col.validity()
            o = col.offset
            if col.kind in BINARY_KINDS:
                keys.append([bytes(col.buf2[col.buf1[o + i]:col.buf1[o + i + 1]]) if v[i] else None for i in range(n)])
            else:  # PRIM values / DICT indices: byte-wise
                raw = col.buf1[o:o + n]
                keys.append([(raw[i].tobytes() if raw.dtype != bool else bytes([int(raw[i])])) if v[i] else None for i in range(n)])
        ids = np.zeros(n, np.uint32)
        for i in range(n):
            k = tuple(col[i] for col in keys)
            if k not in seen:
                seen[k] = len(seen)
                firsts.append(k)
            ids[i] = seen[k]
        ids_out.append(ids)
    G = len(firsts)
    uniq = []
    for c, col in enumerate(batches[0]):
        vals = [f[c] for f in firsts]
        valid = np.array([x is not None for x in vals], bool)
        vb = None if valid.all() else valid
        if col.kind in BINARY_KINDS:
            lens = [0 if x is None else len(x) for x in vals]
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32 if col.kind in VAR32 else np.int64)
            data = np.frombuffer(b"".join(x or b"" for x in vals), np.uint8).copy()
            uniq.append(N(col.kind, G, 0, vb, None, offs, data))
        else:
            dt = col.buf1.dtype
            buf = np.zeros(G, dt)
            for g, x in enumerate(vals):
                if x is not None:
                    buf[g] = np.frombuffer(x, dt)[0] if dt != bool else bool(x[0])
            kids = [dicts[c][1]] if col.kind == K_DICT else []
            uniq.append(N(col.kind, G, 0, vb, None, buf, children=kids, type_id=col.type_id))
    return ids_out, G, uniq


def same_where_valid(a, b):
    """uniques of fixed-width columns: the value under a null is unspecified"""
    if a.kind in (K_PRIM, K_DICT):
        m = a.validity()
        if (a.length, a.null_count) != (b.length, b.null_count) or not np.array_equal(m, b.validity()):
            return False, "shape / validity"
        return (np.array_equal(a.buf1[a.offset:a.offset + a.length][m], b.buf1[b.offset:b.offset + b.length][m]), "values")
    return a.same(b)


def rand_key_batches(rng, shape, n, nbatches, p_null=0.12, pool_size=12, maxlen=4):
    """key columns for the Grouper: few distinct keys so that groups repeat within and across batches"""
    pool = [bytes(rng.integers(97, 100, int(rng.integers(0, maxlen))).astype(np.uint8)) for _ in range(pool_size)]
    dict_values = rand_node(rng, K_UTF8, 5, 0.0)

    def col(kind, nb):
        if kind in BINARY_KINDS:
            items = [pool[rng.integers(0, pool_size)] if rng.random() >= p_null else None for _ in range(nb)]
            lens = [0 if x is None else len(x) for x in items]
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32 if kind in VAR32 else np.int64)
            valid = np.array([x is not None for x in items], bool)
            return N(kind, nb, 0, None if valid.all() else valid, None, offs, np.frombuffer(b"".join(x or b"" for x in items), np.uint8).copy())
        if kind == K_DICT:
            return N(K_DICT, nb, 0, rand_valid(rng, nb, p_null), None, rng.integers(0, 5, nb).astype(np.int8), children=[dict_values], type_id=orc.INT8)
        return N(K_PRIM, nb, 0, rand_valid(rng, nb, p_null), None, rng.integers(0, 4, nb).astype(np.int32), type_id=INT32)
    return [[col(k, n) for k in shape] for _ in range(nbatches)]


GROUP_SHAPES = [[K_UTF8], [K_BINARY], [K_LARGE_BINARY], [K_LARGE_UTF8], [K_DICT], [K_UTF8, K_PRIM], [K_PRIM, K_BINARY, K_DICT]]


def group_cases():
    rng = np.random.default_rng(0x0ff1ce + 4)
    for shape in GROUP_SHAPES:
        for trial in range(4):
            yield rand_key_batches(rng, shape, int(rng.integers(1, 150)), 3, float(rng.choice([0, 0.12])))


# ---- random arrays ------------------------------------------------------------------------------------------------
def rand_valid(rng, n, p_null):
    if p_null == 0:
        return None
    v = rng.random(n) >= p_null
    return v


def rand_node(rng, kind, n, p_null=0.1, off=0, depth=0):
    """a random array of `kind` with n logical rows behind `off` physical ones"""
    m = n + off
    valid = rand_valid(rng, m, p_null)
    if kind == K_PRIM:
        t = [orc.INT8, orc.UINT16, INT32, INT64, orc.DOUBLE, BOOL][rng.integers(0, 6)]
        data = rng.random(m) < 0.5 if t == BOOL else (rng.integers(-100, 100, m) if t != orc.UINT16 else rng.integers(0, 1000, m))
        return N(K_PRIM, n, off, valid, None, np.asarray(data).astype(bool if t == BOOL else NP_OF[t]), type_id=t)
    if kind in BINARY_KINDS:
        lens = rng.integers(0, 12, m)
        if rng.random() < 0.3 and m:
            lens[rng.integers(0, m)] = 300  # a long row: the cooperative copy
        pad = int(rng.integers(0, 5))
        offs = (pad + np.concatenate([[0], np.cumsum(lens)])).astype(np.int32 if kind in VAR32 else np.int64)
        return N(kind, n, off, valid, None, offs, rng.integers(97, 123, int(offs[-1])).astype(np.uint8))
    if kind in FIXED_KINDS:
        w = {K_FSB: int(rng.choice([1, 3, 4, 6, 8, 12, 16, 20])), K_DEC128: 16, K_DEC256: 32}[kind]
        return N(kind, n, off, valid, None, rng.integers(0, 256, m * w).astype(np.uint8), byte_width=w)
    if kind in LIST_KINDS:
        lens = rng.integers(0, 5, m)
        if rng.random() < 0.3 and m:
            lens[rng.integers(0, m)] = 90
        pad = int(rng.integers(0, 3))
        offs = (pad + np.concatenate([[0], np.cumsum(lens)])).astype(np.int32 if kind in VAR32 else np.int64)
        total = int(offs[-1])
        if kind == K_MAP:
            keys = rand_node(rng, K_PRIM, total, 0.0)
            items = rand_node(rng, K_PRIM if depth else K_UTF8, total, p_null)
            child = N(K_STRUCT, total, 0, None, None, children=[keys, items])
        else:
            child = rand_node(rng, [K_PRIM, K_UTF8, K_LIST][rng.integers(0, 2 if depth else 3)], total, p_null, 0, depth + 1)
        return N(kind, n, off, valid, None, offs, children=[child])
    if kind == K_FSL:
        ls = int(rng.integers(1, 4))
        child = rand_node(rng, [K_PRIM, K_UTF8][rng.integers(0, 2)], m * ls, p_null, 0, depth + 1)
        return N(K_FSL, n, off, valid, None, children=[child], list_size=ls)
    if kind == K_STRUCT:
        kids = [rand_node(rng, kk, m, p_null, 0, depth + 1) for kk in ([K_PRIM, K_UTF8, K_FSB] if depth else [K_PRIM, K_UTF8, K_LIST, K_FSB])]
        return N(K_STRUCT, n, off, valid, None, children=kids)
    if kind == K_DICT:
        d = rand_node(rng, [K_PRIM, K_UTF8][rng.integers(0, 2)], 7, 0.0)
        t = [orc.INT8, INT32, orc.UINT16][rng.integers(0, 3)]
        return N(K_DICT, n, off, valid, None, rng.integers(0, 7, m).astype(NP_OF[t]), children=[d], type_id=t)
    if kind == K_NULL:
        return N(K_NULL, n, off)
    raise NotImplementedError(kind)


def rand_indices(rng, n_values, n, p_null, off=0):
    t = [orc.INT8, orc.UINT16, INT32, orc.UINT32, INT64, orc.UINT64][rng.integers(0, 6)]
    hi = max(min(n_values, 120 if t == orc.INT8 else n_values), 1)
    data = rng.integers(0, hi, n + off).astype(NP_OF[t])
    return Arr(t, data, rand_valid(rng, n + off, p_null), off, n)


# ---- committed outputs of the real reference (tests/golden/ref_nested.*, made by tests/golden/make_golden.py) ----------
def load_golden_nested():
    import json
    base = os.path.join(orc.ROOT, "tests", "golden", "ref_nested")
    blob = np.load(base + ".npz")["blob"]
    cases = json.load(open(base + ".json"))["cases"]

    def get(ref):
        if ref is None:
            return None
        off, dt, cnt = ref
        if dt == "bool":
            return unpack(blob[off:off + (cnt + 7) // 8], cnt)
        d = np.dtype(dt)
        return blob[off:off + cnt * d.itemsize].view(d).copy()

    def tree(j):
        return N(j["kind"], j["length"], j["offset"], get(j["valid"]), j["null_count"], get(j["buf1"]), get(j["buf2"]),
                 [tree(c) for c in j["children"]], j["type_id"], j["byte_width"], j["list_size"])
    for e in cases:
        s = e["selection"]
        sel = Arr(s["type"], get(s["data"]), get(s["valid"]), s["offset"], s["length"], s["null_count"])
        yield tree(e["values"]), sel, e["is_filter"], e["option"], (tree(e["result"]) if "result" in e else None), e.get("error")


def load_golden_groups():
    import json
    base = os.path.join(orc.ROOT, "tests", "golden", "ref_nested")
    blob = np.load(base + ".npz")["blob"]
    doc = json.load(open(base + ".json"))

    def get(ref):
        if ref is None:
            return None
        off, dt, cnt = ref
        if dt == "bool":
            return unpack(blob[off:off + (cnt + 7) // 8], cnt)
        d = np.dtype(dt)
        return blob[off:off + cnt * d.itemsize].view(d).copy()

    def tree(j):
        return N(j["kind"], j["length"], j["offset"], get(j["valid"]), j["null_count"], get(j["buf1"]), get(j["buf2"]),
                 [tree(c) for c in j["children"]], j["type_id"], j["byte_width"], j["list_size"])
    for e in doc["group_cases"]:
        yield [[tree(c) for c in cols] for cols in e["batches"]], [get(i) for i in e["ids"]], e["num_groups"], [tree(u) for u in e["uniques"]]


def check_groups_against_golden(group_fn):
    bad = []
    for i, (batches, ids, ng, uniq) in enumerate(load_golden_groups()):
        got = group_fn(batches)
        if got[1] != ng or not all(np.array_equal(x, y) for x, y in zip(got[0], ids)):
            bad.append("group case %d: ids / group count differ" % i)
            continue
        for c, (ua, ub) in enumerate(zip(uniq, got[2])):
            ok, msg = same_where_valid(ua, ub)
            if not ok:
                bad.append("group case %d column %d: %s" % (i, c, msg))
    return bad


def check_against_golden(impl):
    """-> list of failure messages for `impl.select` over every committed case"""
    bad = []
    for i, (v, sel, is_filter, option, want, err) in enumerate(load_golden_nested()):
        what = "case %d %s %s" % (i, KIND_NAMES[v.kind], "filter" if is_filter else "take")
        try:
            got = impl.select(v, sel, is_filter, option)
        except orc.StatusError as e:
            if err is None or [e.code, e.msg] != err:
                bad.append("%s: raised %r, expected %r" % (what, (e.code, e.msg), err))
            continue
        if err is not None:
            bad.append("%s: expected error %r" % (what, err))
            continue
        ok, msg = want.same(got)
        if not ok:
            bad.append("%s: %s" % (what, msg))
    return bad
