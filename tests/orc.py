"""ctypes access to the CPU oracle (oracle/liboracle.so, `orc_*`) and -- only where
oracle/_ref/ has been built -- to the real reference (`ref_*`, oracle/_ref/libref_driver.so).

TEST INFRASTRUCTURE.  Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only; nothing under arrow-1_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BOOL, UINT8, INT8, UINT16, INT16, UINT32, INT32, UINT64, INT64, FLOAT, DOUBLE = 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12
NP_OF = {UINT8: np.uint8, INT8: np.int8, UINT16: np.uint16, INT16: np.int16, UINT32: np.uint32,
         INT32: np.int32, UINT64: np.uint64, INT64: np.int64, FLOAT: np.float32, DOUBLE: np.float64}
ID_OF = {np.dtype(v): k for k, v in NP_OF.items()}
NUMERIC = [UINT8, INT8, UINT16, INT16, UINT32, INT32, UINT64, INT64, FLOAT, DOUBLE]
INTS = [UINT8, INT8, UINT16, INT16, UINT32, INT32, UINT64, INT64]
DROP, EMIT_NULL = 0, 1
ASC, DESC = 0, 1
EQ, NE, GT, GE, LT, LE = range(6)
INVERT, AND, AND_NOT, OR, XOR, AND_KLEENE, AND_NOT_KLEENE, OR_KLEENE = range(8)
SCALAR_NONE, SCALAR_FALSE, SCALAR_TRUE, SCALAR_NULL = -1, 0, 1, 2
ADD, SUB, MUL, DIV, ADD_C, SUB_C, MUL_C, DIV_C = range(8)
OK, KEY_ERROR, TYPE_ERROR, INVALID, INDEX_ERROR, NOT_IMPLEMENTED = 0, 2, 3, 4, 7, 10


def bit_width(t):
    return 1 if t == BOOL else np.dtype(NP_OF[t]).itemsize * 8


def pack(bits):
    """bool array -> Arrow LSB-first bitmap bytes"""
    return np.packbits(np.asarray(bits, dtype=bool), bitorder="little")


def unpack(buf, n, offset=0):
    return np.unpackbits(np.asarray(buf, dtype=np.uint8), bitorder="little")[offset:offset + n].astype(bool)


class CArr(C.Structure):
    _fields_ = [("values", C.c_void_p), ("validity", C.c_void_p), ("length", C.c_int64),
                ("offset", C.c_int64), ("type_id", C.c_int32), ("null_count", C.c_int64)]


class CBin(C.Structure):
    _fields_ = [("offsets", C.c_void_p), ("data", C.c_void_p), ("validity", C.c_void_p), ("length", C.c_int64),
                ("offset", C.c_int64), ("null_count", C.c_int64)]


class COut(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("has_validity", C.c_int32)]


class Arr:
    """A primitive Arrow array over numpy buffers.

    `data`: the PHYSICAL values buffer as an unpacked numpy array (bool array for BOOL),
    including `offset` leading elements; `valid`: same-length bool array or None (no
    validity buffer).  null_count None -> computed; pass -1 for kUnknownNullCount."""

    def __init__(self, type_id, data, valid=None, offset=0, length=None, null_count=None):
        self.type_id = type_id
        data = np.asarray(data)
        self.data = data.astype(bool) if type_id == BOOL else np.ascontiguousarray(data, dtype=NP_OF[type_id])
        self.valid = None if valid is None else np.asarray(valid, dtype=bool)
        self.offset = offset
        self.length = len(self.data) - offset if length is None else length
        if self.valid is None:
            self.null_count = 0
        elif null_count is None:
            self.null_count = int((~self.valid[offset:offset + self.length]).sum())
        else:
            self.null_count = null_count
        self.values_buf = pack(self.data) if type_id == BOOL else self.data
        # pad so that word-at-a-time readers never run off the end
        self.values_buf = np.concatenate([self.values_buf.view(np.uint8).ravel(), np.zeros(16, np.uint8)])
        self.valid_buf = None if self.valid is None else np.concatenate([pack(self.valid), np.zeros(16, np.uint8)])

    def c(self):
        return CArr(self.values_buf.ctypes.data, None if self.valid_buf is None else self.valid_buf.ctypes.data,
                    self.length, self.offset, self.type_id, self.null_count)

    # logical views
    def values(self):
        return self.data[self.offset:self.offset + self.length]

    def validity(self):
        if self.valid is None:
            return np.ones(self.length, bool)
        return self.valid[self.offset:self.offset + self.length]


class CAgg(C.Structure):
    _fields_ = [("valid_count", C.c_int64), ("sum", C.c_uint64), ("min", C.c_uint64), ("max", C.c_uint64)]


def agg_to_dict(type_id, valid_count, s, mn, mx):
    def conv(bits):
        if type_id in (FLOAT, DOUBLE):
            return float(np.array([bits], np.uint64).view(np.float64)[0])
        if type_id in (INT8, INT16, INT32, INT64):
            return int(np.array([bits], np.uint64).view(np.int64)[0])
        return int(bits)
    return {"valid_count": int(valid_count), "sum": conv(s), "min": conv(mn), "max": conv(mx)}


class BinArr:
    """A BinaryArray (32-bit offsets) for the test APIs: items = list of bytes / None."""

    def __init__(self, items, offset=0, length=None, null_count=None, pad=0):
        items = [None] * 0 + list(items)
        self.items = items
        lens = [0 if x is None else len(x) for x in items]
        self.offsets = np.concatenate([[pad], pad + np.cumsum(lens)]).astype(np.int32)
        self.data = np.frombuffer(b"\0" * pad + b"".join(x or b"" for x in items) + b"\0" * 8, np.uint8).copy()
        valid = np.array([x is not None for x in items], bool)
        self.valid = None if valid.all() else valid
        self.vbits = None if self.valid is None else np.concatenate([pack(valid), np.zeros(8, np.uint8)])
        self.offset = offset
        self.length = len(items) - offset if length is None else length
        nc = int((~valid[offset:offset + self.length]).sum())
        self.null_count = nc if null_count is None else null_count

    def c(self):
        s = CBin()
        s.offsets = self.offsets.ctypes.data_as(C.c_void_p)
        s.data = self.data.ctypes.data_as(C.c_void_p)
        s.validity = None if self.vbits is None else self.vbits.ctypes.data_as(C.c_void_p)
        s.length, s.offset, s.null_count = self.length, self.offset, self.null_count if self.vbits is not None else 0
        return s

    def logical(self):
        return self.items[self.offset:self.offset + self.length]


class BinResult:
    def __init__(self, offsets, data, validity_bits, out):
        self.length, self.null_count, self.has_validity = out.length, out.null_count, bool(out.has_validity)
        self.offsets = offsets[:self.length + 1].copy()
        self.data = data[:int(self.offsets[-1]) if self.length else 0].copy()
        self.validity_bits = validity_bits[:(self.length + 7) // 8].copy()

    def validity(self):
        return unpack(self.validity_bits, self.length)

    def logical(self):
        v = self.validity()
        return [bytes(self.data[self.offsets[i]:self.offsets[i + 1]]) if v[i] else None for i in range(self.length)]

    def same(self, other):
        return ((self.length, self.null_count, self.has_validity) == (other.length, other.null_count, other.has_validity) and
                np.array_equal(self.offsets, other.offsets) and np.array_equal(self.data, other.data) and
                np.array_equal(self.validity(), other.validity()))


class Result:
    def __init__(self, values, validity_bits, out, type_id):
        self.values, self.validity_bits, self.type_id = values, validity_bits, type_id
        self.length, self.null_count, self.has_validity = out.length, out.null_count, bool(out.has_validity)

    def validity(self):
        return unpack(self.validity_bits, self.length)

    def same(self, other, check_validity_bytes=True):
        if (self.length, self.null_count, self.has_validity) != (other.length, other.null_count, other.has_validity):
            return False
        if not np.array_equal(self.values, other.values):
            return False
        if check_validity_bytes and self.has_validity and not np.array_equal(self.validity_bits, other.validity_bits):
            return False
        return True

    def same_valid(self, other):
        """like same(), but value bytes are compared only where the slot is valid: for outputs whose
        null slots the reference leaves unspecified (the memo table's null entry, vector_hash.cc)"""
        if (self.length, self.null_count, self.has_validity) != (other.length, other.null_count, other.has_validity):
            return False
        v = self.validity() if self.has_validity else np.ones(self.length, bool)
        w = other.validity() if other.has_validity else np.ones(other.length, bool)
        if not np.array_equal(v, w):
            return False
        a, b = np.asarray(self.values)[:self.length][v], np.asarray(other.values)[:other.length][v]
        return np.array_equal(a.view(np.uint8) if a.dtype != bool else a, b.view(np.uint8) if b.dtype != bool else b)

    def __repr__(self):
        return "Result(len=%d nulls=%d has_validity=%s values=%s validity=%s)" % (
            self.length, self.null_count, self.has_validity, self.values[:16], self.validity()[:16] if self.length else [])


class StatusError(Exception):
    def __init__(self, code, msg):
        super().__init__("status %d: %s" % (code, msg))
        self.code, self.msg = code, msg


POWER, POWER_CHECKED, NEGATE, NEGATE_CHECKED = 8, 9, 10, 11


def build_oracle():
    """gcc-compile oracle/oracle.c -> oracle/liboracle.so (seconds)."""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "oracle.c")
    hdr = os.path.join(ROOT, "oracle", "oracle.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-misleading-indentation",
                               "-o", so, src, "-lm"])
    return so


class Api:
    """The orc_* (prefix 'orc') or ref_* (prefix 'ref') entry points with numpy in/out."""

    def __init__(self, prefix="orc"):
        self.prefix = prefix
        if prefix == "orc":
            self.lib = C.CDLL(build_oracle())
        else:
            path = os.path.join(ROOT, "oracle", "_ref", "libref_driver.so")
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            self.lib = C.CDLL(path)
        f = self._f
        f("last_error").restype = C.c_char_p
        f("filter_output_size").restype = C.c_int64
        f("filter_groupby_sum").restype = C.c_int64
        f("grouper_new").restype = C.c_void_p
        f("grouper_num_groups").restype = C.c_uint32
        f("grouper_num_groups").argtypes = [C.c_void_p]
        f("grouper_free").argtypes = [C.c_void_p]

    def _f(self, name):
        return getattr(self.lib, self.prefix + "_" + name)

    def _check(self, rc):
        if rc != 0:
            raise StatusError(rc, self._f("last_error")().decode())

    def filter_output_size(self, filt, null_sel=DROP):
        c = filt.c()
        return self._f("filter_output_size")(C.byref(c), null_sel)

    def filter(self, values, filt, null_sel=DROP):
        bw = bit_width(values.type_id)
        cap = max(filt.length, values.length) + 64
        out_v = np.zeros(cap * max(bw, 8) // 8 + 16, np.uint8)
        out_b = np.zeros(cap // 8 + 16, np.uint8)
        out = COut()
        cv, cf = values.c(), filt.c()
        self._check(self._f("filter")(C.byref(cv), C.byref(cf), null_sel, out_v.ctypes.data_as(C.c_void_p),
                                      out_b.ctypes.data_as(C.c_void_p), C.byref(out)))
        return self._result(values.type_id, out_v, out_b, out)

    def _result(self, type_id, out_v, out_b, out):
        n = out.length
        if type_id == BOOL:
            vals = unpack(out_v, n)
        else:
            vals = out_v[:n * bit_width(type_id) // 8].view(NP_OF[type_id]).copy()
        return Result(vals, out_b[:(n + 7) // 8].copy(), out, type_id)

    def take_indices(self, filt, null_sel=DROP):
        cap = filt.length + 64
        out_v = np.zeros(cap * 4 + 16, np.uint8)
        out_b = np.zeros(cap // 8 + 16, np.uint8)
        out, w = COut(), C.c_int(0)
        cf = filt.c()
        self._check(self._f("take_indices")(C.byref(cf), null_sel, out_v.ctypes.data_as(C.c_void_p),
                                            out_b.ctypes.data_as(C.c_void_p), C.byref(w), C.byref(out)))
        return self._result(UINT16 if w.value == 16 else UINT32, out_v, out_b, out)

    def take(self, values, indices, boundscheck=True):
        cap = indices.length + 64
        out_v = np.zeros(cap * 8 + 16, np.uint8)
        out_b = np.zeros(cap // 8 + 16, np.uint8)
        out = COut()
        cv, ci = values.c(), indices.c()
        self._check(self._f("take")(C.byref(cv), C.byref(ci), int(boundscheck), out_v.ctypes.data_as(C.c_void_p),
                                    out_b.ctypes.data_as(C.c_void_p), C.byref(out)))
        return self._result(values.type_id, out_v, out_b, out)

    def sort_indices(self, chunks, order=ASC):
        if isinstance(chunks, Arr):
            chunks = [chunks]
        arr = (CArr * max(len(chunks), 1))(*[c.c() for c in chunks])
        total = sum(c.length for c in chunks)
        out = np.zeros(total + 1, np.uint64)
        self._check(self._f("sort_indices")(arr, len(chunks), order, out.ctypes.data_as(C.c_void_p)))
        return out[:total]

    # ---- variable-width Take / Filter ----
    def _bin_call(self, fn, values, sel, flag, n_out_max):
        total = int(values.offsets[-1]) + 16
        out_o = np.zeros(n_out_max + 2, np.int32)
        out_d = np.zeros(max(total * max(1, (n_out_max + values.length) // max(values.length, 1)), 16) + 16, np.uint8)
        out_b = np.zeros(n_out_max // 8 + 16, np.uint8)
        out, cv, cs = COut(), values.c(), sel.c()
        self._check(self._f(fn)(C.byref(cv), C.byref(cs), flag, out_o.ctypes.data_as(C.c_void_p), out_d.ctypes.data_as(C.c_void_p),
                                out_b.ctypes.data_as(C.c_void_p), C.byref(out)))
        return BinResult(out_o, out_d, out_b, out)

    def binary_concatenate(self, chunks):
        rows = sum(c.length for c in chunks)
        total = sum(int(c.offsets[-1]) for c in chunks) + 16
        out_o, out_d, out_b = np.zeros(rows + 2, np.int32), np.zeros(total, np.uint8), np.zeros(rows // 8 + 16, np.uint8)
        out = COut()
        arr = (CBin * len(chunks))(*[c.c() for c in chunks])
        self._check(self._f("binary_concatenate")(arr, len(chunks), out_o.ctypes.data_as(C.c_void_p), out_d.ctypes.data_as(C.c_void_p),
                                                  out_b.ctypes.data_as(C.c_void_p), C.byref(out)))
        return BinResult(out_o, out_d, out_b, out)

    def binary_take(self, values, indices, boundscheck=True):
        return self._bin_call("binary_take", values, indices, int(boundscheck), indices.length)

    def binary_filter(self, values, filt, null_sel=DROP):
        return self._bin_call("binary_filter", values, filt, null_sel, filt.length)

    # ---- unique / value_counts / dictionary_encode ----
    def unique(self, a):
        out_v = np.zeros(a.length * 8 + 16, np.uint8)
        out_b = np.zeros(a.length // 8 + 16, np.uint8)
        out, c = COut(), a.c()
        self._check(self._f("unique")(C.byref(c), out_v.ctypes.data_as(C.c_void_p), out_b.ctypes.data_as(C.c_void_p), C.byref(out)))
        return self._result(a.type_id, out_v, out_b, out)

    def value_counts(self, a):
        out_v = np.zeros(a.length * 8 + 16, np.uint8)
        out_b = np.zeros(a.length // 8 + 16, np.uint8)
        out_c = np.zeros(a.length + 2, np.int64)
        out, c = COut(), a.c()
        self._check(self._f("value_counts")(C.byref(c), out_v.ctypes.data_as(C.c_void_p), out_b.ctypes.data_as(C.c_void_p),
                                            out_c.ctypes.data_as(C.c_void_p), C.byref(out)))
        return self._result(a.type_id, out_v, out_b, out), out_c[:out.length].copy()

    def dictionary_encode(self, a, null_encoding=1):
        """-> (indices Result int32, dictionary Result); null_encoding 0 = ENCODE, 1 = MASK"""
        out_i = np.zeros(a.length * 4 + 16, np.uint8)
        out_ib = np.zeros(a.length // 8 + 16, np.uint8)
        out_v = np.zeros(a.length * 8 + 16, np.uint8)
        out_b = np.zeros(a.length // 8 + 16, np.uint8)
        om, dm, c = COut(), COut(), a.c()
        self._check(self._f("dictionary_encode")(C.byref(c), null_encoding, out_i.ctypes.data_as(C.c_void_p),
                                                 out_ib.ctypes.data_as(C.c_void_p), C.byref(om), out_v.ctypes.data_as(C.c_void_p),
                                                 out_b.ctypes.data_as(C.c_void_p), C.byref(dm)))
        return self._result(INT32, out_i, out_ib, om), self._result(a.type_id, out_v, out_b, dm)

    def hash_min_max(self, values, ids, num_groups, null_handling=0):
        """-> (mins Result, maxes Result) sharing one validity"""
        ids = np.ascontiguousarray(ids, np.uint32)
        w = max(bit_width(values.type_id), 8) // 8
        out_mn = np.zeros(num_groups * w + 16, np.uint8)
        out_mx = np.zeros(num_groups * w + 16, np.uint8)
        out_b = np.zeros(num_groups // 8 + 16, np.uint8)
        out = COut()
        cv = values.c()
        self._check(self._f("hash_min_max")(C.byref(cv), ids.ctypes.data_as(C.c_void_p), C.c_int64(num_groups), null_handling,
                                            out_mn.ctypes.data_as(C.c_void_p), out_mx.ctypes.data_as(C.c_void_p),
                                            out_b.ctypes.data_as(C.c_void_p), C.byref(out)))
        return self._result(values.type_id, out_mn, out_b, out), self._result(values.type_id, out_mx, out_b, out)

    def sort_indices_multi(self, keys, orders):
        """keys: one Arr or list of chunk Arrs per sort key; orders: ASC/DESC per key."""
        keys = [[k] if isinstance(k, Arr) else list(k) for k in keys]
        flat = [c for k in keys for c in k]
        arr = (CArr * max(len(flat), 1))(*[c.c() for c in flat])
        nch = (C.c_int * max(len(keys), 1))(*[len(k) for k in keys])
        ords = (C.c_int * max(len(keys), 1))(*[int(o) for o in orders])
        total = sum(c.length for c in keys[0]) if keys else 0
        out = np.zeros(total + 1, np.uint64)
        self._check(self._f("sort_indices_multi")(arr, nch, ords, len(keys), out.ctypes.data_as(C.c_void_p)))
        return out[:total]

    def concatenate(self, chunks):
        total = sum(c.length for c in chunks)
        out_v = np.zeros(total * 8 + 16, np.uint8)
        out_b = np.zeros(total // 8 + 16, np.uint8)
        out = COut()
        arr = (CArr * len(chunks))(*[c.c() for c in chunks])
        self._check(self._f("concatenate")(arr, len(chunks), out_v.ctypes.data_as(C.c_void_p),
                                           out_b.ctypes.data_as(C.c_void_p), C.byref(out)))
        return self._result(chunks[0].type_id, out_v, out_b, out)

    def group_ids(self, key_batches):
        """key_batches: list of batches, each a list of key-column Arr.  Returns
        (ids per batch, num_groups, uniques per key column as (values, validity, null_count))."""
        nkeys = len(key_batches[0])
        types = (C.c_int * nkeys)(*[k.type_id for k in key_batches[0]])
        g = self._f("grouper_new")(nkeys, types)
        assert g
        try:
            ids = []
            for batch in key_batches:
                arr = (CArr * nkeys)(*[k.c() for k in batch])
                out = np.zeros(batch[0].length + 1, np.uint32)
                self._check(self._f("grouper_consume")(C.c_void_p(g), arr, out.ctypes.data_as(C.c_void_p)))
                ids.append(out[:batch[0].length])
            ng = self._f("grouper_num_groups")(C.c_void_p(g))
            uniq = []
            for k in range(nkeys):
                t = key_batches[0][k].type_id
                ov = np.zeros(ng * 8 + 16, np.uint8)
                ob = np.zeros(ng // 8 + 16, np.uint8)
                nc = C.c_int64(0)
                self._check(self._f("grouper_uniques")(C.c_void_p(g), k, ov.ctypes.data_as(C.c_void_p),
                                                       ob.ctypes.data_as(C.c_void_p), C.byref(nc)))
                vals = unpack(ov, ng) if t == BOOL else ov[:ng * bit_width(t) // 8].view(NP_OF[t]).copy()
                uniq.append((vals, unpack(ob, ng), nc.value))
            return ids, ng, uniq
        finally:
            self._f("grouper_free")(C.c_void_p(g))

    def make_groupings(self, ids, num_groups):
        """Grouper::MakeGroupings -> (offsets int32[num_groups + 1], indices int32[n])"""
        ids = np.ascontiguousarray(ids, np.uint32)
        off = np.zeros(num_groups + 1, np.int32)
        idx = np.zeros(max(len(ids), 1), np.int32)
        self._check(self._f("make_groupings")(ids.ctypes.data_as(C.c_void_p), C.c_int64(len(ids)), C.c_uint32(num_groups),
                                              off.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p)))
        return off, idx[:len(ids)]

    def hash_aggregate(self, func, values, ids, num_groups, count_mode=0):
        """func: 'sum' | 'count' | 'mean'.  Returns Result with 8-byte values viewed per out type."""
        fid = {"sum": 0, "count": 1, "mean": 2}[func]
        ids = np.ascontiguousarray(ids, np.uint32)
        ov = np.zeros(num_groups * 8 + 16, np.uint8)
        ob = np.zeros(num_groups // 8 + 16, np.uint8)
        out = COut()
        cv = values.c()
        self._check(self._f("hash_aggregate")(fid, C.byref(cv), ids.ctypes.data_as(C.c_void_p), C.c_int64(num_groups),
                                              count_mode, ov.ctypes.data_as(C.c_void_p), ob.ctypes.data_as(C.c_void_p),
                                              C.byref(out)))
        t = values.type_id
        if func == "count":
            ot = INT64
        elif func == "mean" or t in (FLOAT, DOUBLE):
            ot = DOUBLE
        elif t in (INT8, INT16, INT32, INT64):
            ot = INT64
        else:
            ot = UINT64
        return self._result(ot, ov, ob, out)

    def _binary(self, name, op, lhs, rhs, out_type):
        """lhs / rhs: Arr or a python/numpy scalar (taken in the array operand's type)."""
        arr = lhs if isinstance(lhs, Arr) else rhs
        t, n = arr.type_id, arr.length

        def operand(x):
            if isinstance(x, Arr):
                c = x.c()
                return C.byref(c), None, c
            s = np.array([x], dtype=np.uint8 if t == BOOL else NP_OF[t])
            return None, s.ctypes.data_as(C.c_void_p), s

        la, ls, keep1 = operand(lhs)
        ra, rs, keep2 = operand(rhs)
        if isinstance(lhs, Arr) and isinstance(rhs, Arr) and lhs.type_id != rhs.type_id:
            t = self.common_numeric(lhs.type_id, rhs.type_id)  # e3: implicit promotion
        ot = t if out_type is None else out_type
        ov = np.zeros((n + 64) * 8 + 16, np.uint8)
        ob = np.zeros(n // 8 + 16, np.uint8)
        out = COut()
        self._check(self._f(name)(op, la, ls, ra, rs, ov.ctypes.data_as(C.c_void_p), ob.ctypes.data_as(C.c_void_p),
                                  C.byref(out)))
        return self._result(ot, ov, ob, out)

    def common_numeric(self, type_a, type_b):
        """CommonNumeric (codegen_internal.cc:220-268) -> type id, -1 if there is none"""
        return int(self._f("common_numeric")(int(type_a), int(type_b)))

    def cast(self, a, to_type, safe=True):
        """number -> number cast: the converted values (the validity is the input's)"""
        out_v = np.zeros(a.length * 8 + 16, np.uint8)
        c = a.c()
        self._check(self._f("cast_numeric")(C.byref(c), int(to_type), int(bool(safe)), out_v.ctypes.data_as(C.c_void_p)))
        return out_v[:a.length * bit_width(to_type) // 8].view(NP_OF[to_type]).copy()

    def compare(self, op, lhs, rhs):
        return self._binary("compare", op, lhs, rhs, BOOL)

    def arith(self, op, lhs, rhs=None):
        """op 0-7 add ... divide_checked, 8 / 9 power[_checked], 10 / 11 negate[_checked] (unary: rhs unused)"""
        if op >= NEGATE:
            ov = np.zeros((lhs.length + 64) * 8 + 16, np.uint8)
            ob = np.zeros(lhs.length // 8 + 16, np.uint8)
            out, c = COut(), lhs.c()
            self._check(self._f("arith")(op, C.byref(c), None, None, None, ov.ctypes.data_as(C.c_void_p), ob.ctypes.data_as(C.c_void_p),
                                         C.byref(out)))
            return self._result(lhs.type_id, ov, ob, out)
        return self._binary("arith", op, lhs, rhs, None)

    # ---- mask-building scalars ----
    def boolean(self, op, lhs, rhs=None):
        """lhs / rhs: Arr(BOOL) or a scalar state: False, True or None (= null scalar)."""
        def operand(x):
            if isinstance(x, Arr):
                c = x.c()
                return C.byref(c), SCALAR_NONE, c
            return None, (SCALAR_NULL if x is None else int(bool(x))), None
        arr = lhs if isinstance(lhs, Arr) else rhs
        n = arr.length
        la, ls, k1 = operand(lhs)
        ra, rs, k2 = operand(rhs) if op != INVERT else (None, SCALAR_TRUE, None)
        ov = np.zeros(n // 8 + 16, np.uint8)
        ob = np.zeros(n // 8 + 16, np.uint8)
        out = COut()
        self._check(self._f("boolean")(op, la, ls, ra, rs, ov.ctypes.data_as(C.c_void_p),
                                       ob.ctypes.data_as(C.c_void_p), C.byref(out)))
        return self._result(BOOL, ov, ob, out)

    def _validity_test(self, func, a):
        ov = np.zeros(a.length // 8 + 16, np.uint8)
        out = COut()
        c = a.c()
        self._check(self._f("validity_test")(func, C.byref(c), ov.ctypes.data_as(C.c_void_p), C.byref(out)))
        return self._result(BOOL, ov, np.zeros(a.length // 8 + 16, np.uint8), out)

    def is_valid(self, a):
        return self._validity_test(0, a)

    def is_null(self, a):
        return self._validity_test(1, a)

    def is_nan(self, a):
        ov = np.zeros(a.length // 8 + 16, np.uint8)
        ob = np.zeros(a.length // 8 + 16, np.uint8)
        out = COut()
        c = a.c()
        self._check(self._f("is_nan")(C.byref(c), ov.ctypes.data_as(C.c_void_p), ob.ctypes.data_as(C.c_void_p),
                                      C.byref(out)))
        return self._result(BOOL, ov, ob, out)

    def fill_null(self, a, fill):
        """fill: a python/numpy scalar of a's type, or None for a null scalar."""
        n = a.length
        ov = np.zeros((n + 64) * 8 + 16, np.uint8)
        ob = np.zeros(n // 8 + 16, np.uint8)
        out = COut()
        c = a.c()
        keep = None if fill is None else np.array([fill], dtype=np.uint8 if a.type_id == BOOL else NP_OF[a.type_id])
        fp = None if fill is None else keep.ctypes.data_as(C.c_void_p)
        self._check(self._f("fill_null")(C.byref(c), fp, ov.ctypes.data_as(C.c_void_p), ob.ctypes.data_as(C.c_void_p),
                                         C.byref(out)))
        return self._result(a.type_id, ov, ob, out)

    # ---- scalar aggregates ----
    def aggregate(self, a):
        """-> dict(valid_count, sum, min, max) in the accumulator family of a's type
        (python int for integers / boolean, float for floating point)."""
        out = CAgg()
        c = a.c()
        self._check(self._f("aggregate")(C.byref(c), C.byref(out)))
        return agg_to_dict(a.type_id, out.valid_count, out.sum, out.min, out.max)

    def filter_groupby_sum(self, values, keys, mask):
        n = values.length
        sums = np.zeros(n + 1, np.uint64)
        counts = np.zeros(n + 1, np.int64)
        okeys = np.zeros((n + 1) * 8, np.uint8)
        okv = np.zeros(n // 8 + 16, np.uint8)
        cv, ck, cm = values.c(), keys.c(), mask.c()
        ng = self._f("filter_groupby_sum")(C.byref(cv), C.byref(ck), C.byref(cm), sums.ctypes.data_as(C.c_void_p),
                                           counts.ctypes.data_as(C.c_void_p), okeys.ctypes.data_as(C.c_void_p),
                                           okv.ctypes.data_as(C.c_void_p))
        if ng < 0:
            raise StatusError(-ng, self._f("last_error")().decode())
        kt = keys.type_id
        acc = np.float64 if values.type_id in (FLOAT, DOUBLE) else (
            np.int64 if values.type_id in (INT8, INT16, INT32, INT64) else np.uint64)
        okey = unpack(okeys, ng) if kt == BOOL else okeys[:ng * bit_width(kt) // 8].view(NP_OF[kt]).copy()
        return (sums[:ng].view(acc).copy(), counts[:ng].copy(), okey, unpack(okv, ng))


def have_ref():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_driver.so"))


def random_arr(rng, type_id, n, null_prob=0.0, offset=0, lo=None, hi=None, null_count=None, force_validity=False):
    """Random array in the spirit of RandomArrayGenerator (arrow/testing/random.h:41-)."""
    m = n + offset
    if type_id == BOOL:
        data = rng.random(m) < (0.5 if hi is None else hi)
    elif type_id in (FLOAT, DOUBLE):
        data = rng.uniform(-100 if lo is None else lo, 100 if hi is None else hi, m)
    else:
        info = np.iinfo(NP_OF[type_id])
        a = max(info.min, -100 if lo is None else lo)
        b = min(info.max, 100 if hi is None else hi)
        data = rng.integers(a, b, m, endpoint=True, dtype=np.uint64) if b > 2**63 - 1 else rng.integers(a, b, m, endpoint=True)
    valid = None
    if null_prob > 0 or force_validity:
        valid = rng.random(m) >= null_prob
    return Arr(type_id, data, valid, offset=offset, length=n, null_count=null_count)
