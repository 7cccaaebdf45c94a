"""Device-side Arrow arrays over torch (ROCm) storage, and the per-stream Context.

torch is plumbing here: it owns HBM allocations and the HIP stream the kernels run on."""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._abi import BOOL, CArray, CBinaryArray, check, load_library

NP_OF = {_abi.UINT8: np.uint8, _abi.INT8: np.int8, _abi.UINT16: np.uint16, _abi.INT16: np.int16,
         _abi.UINT32: np.uint32, _abi.INT32: np.int32, _abi.UINT64: np.uint64, _abi.INT64: np.int64,
         _abi.FLOAT: np.float32, _abi.DOUBLE: np.float64}


def bit_width(type_id):
    return 1 if type_id == BOOL else np.dtype(NP_OF[type_id]).itemsize * 8


def _pad64(n):
    return ((int(n) + 63) // 64) * 64 + 64


def alloc_bytes(nbytes, device, zero=False):
    """HBM buffer padded like arrow::MemoryPool (64-byte multiples), always 8-byte readable."""
    f = torch.zeros if zero else torch.empty
    return f(_pad64(nbytes), dtype=torch.uint8, device=device)


class Context:
    """One amdarrow_ctx bound to torch's current stream on `device`."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("amdarrow needs a ROCm GPU: there is no CPU path")
        self.lib = load_library()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            h = C.c_void_p()
            check(self.lib, self.lib.amdarrow_ctx_create(self.device.index, C.c_void_p(stream), 0, C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            self.lib.amdarrow_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBinaryArray:
    """A BinaryArray / StringArray (32-bit offsets) in HBM: offsets int32[offset + length + 1], data bytes,
    optional validity bitmap -- buffers validity | offsets | data of array/array_binary.h."""

    def __init__(self, offsets, data, validity, length, offset=0, null_count=0):
        self.offsets, self.data, self.validity = offsets, data, validity
        self.length, self.offset = int(length), int(offset)
        self.null_count = 0 if validity is None else int(null_count)

    @staticmethod
    def from_items(items, offset=0, length=None, null_count=None, device="cuda"):
        """items: list of bytes / None (None = null slot)"""
        lens = [0 if x is None else len(x) for x in items]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        raw = np.frombuffer(b"".join(x or b"" for x in items) + b"\0" * 8, np.uint8)
        o = alloc_bytes(off.nbytes, device, zero=True)
        o[:off.nbytes] = torch.from_numpy(off.view(np.uint8).copy())
        d = alloc_bytes(len(raw), device, zero=True)
        d[:len(raw)] = torch.from_numpy(raw.copy())
        valid = np.array([x is not None for x in items], bool)
        n = len(items) - offset if length is None else length
        v = None
        if not valid.all():
            bits = np.packbits(valid, bitorder="little")
            v = alloc_bytes(len(bits), device, zero=True)
            v[:len(bits)] = torch.from_numpy(bits.copy())
        nc = int((~valid[offset:offset + n]).sum()) if null_count is None else null_count
        return DeviceBinaryArray(o, d, v, n, offset, nc)

    def c(self):
        s = CBinaryArray()
        s.offsets, s.data = self.offsets.data_ptr(), self.data.data_ptr()
        s.validity = None if self.validity is None else self.validity.data_ptr()
        s.length, s.offset, s.null_count = self.length, self.offset, self.null_count
        return s

    def to_items(self):
        off = self.offsets[:(self.offset + self.length + 1) * 4].cpu().numpy().view(np.int32)[self.offset:]
        end = int(off[-1]) if self.length else 0
        data = self.data[:end].cpu().numpy() if end else np.zeros(0, np.uint8)
        if self.validity is None:
            valid = np.ones(self.length, bool)
        else:
            valid = np.unpackbits(self.validity[:(self.offset + self.length + 7) // 8].cpu().numpy(),
                                  bitorder="little")[self.offset:self.offset + self.length].astype(bool)
        return [bytes(data[off[i]:off[i + 1]]) if valid[i] else None for i in range(self.length)]


class DeviceArray:
    """ArrayData{validity, values, offset, length, null_count} with buffers in HBM."""

    def __init__(self, type_id, values, validity, length, offset=0, null_count=0):
        self.type_id, self.values, self.validity = type_id, values, validity
        self.length, self.offset = int(length), int(offset)
        self.null_count = 0 if validity is None else int(null_count)

    @staticmethod
    def from_numpy(type_id, data, valid=None, offset=0, length=None, null_count=None, device="cuda"):
        """data: the physical values buffer as an unpacked numpy array (bool array for BOOL)."""
        data = np.asarray(data)
        n = (len(data) - offset) if length is None else length
        if type_id == BOOL:
            raw = np.packbits(data.astype(bool), bitorder="little")
        else:
            raw = np.ascontiguousarray(data, dtype=NP_OF[type_id]).view(np.uint8).ravel()
        vals = alloc_bytes(len(raw), device, zero=True)
        vals[:len(raw)] = torch.from_numpy(raw.copy())
        vbuf = None
        if valid is not None:
            valid = np.asarray(valid, dtype=bool)
            vraw = np.packbits(valid, bitorder="little")
            vbuf = alloc_bytes(len(vraw), device, zero=True)
            vbuf[:len(vraw)] = torch.from_numpy(vraw)
            if null_count is None:
                null_count = int((~valid[offset:offset + n]).sum())
        return DeviceArray(type_id, vals, vbuf, n, offset, 0 if null_count is None else null_count)

    def c(self):
        return CArray(self.values.data_ptr(), None if self.validity is None else self.validity.data_ptr(),
                      self.length, self.offset, self.type_id, self.null_count)

    def slice(self, offset, length):
        return DeviceArray(self.type_id, self.values, self.validity, length, self.offset + offset,
                           -1 if self.validity is not None else 0)

    # host views (for tests)
    def values_numpy(self):
        raw = self.values.cpu().numpy()
        if self.type_id == BOOL:
            return np.unpackbits(raw, bitorder="little")[self.offset:self.offset + self.length].astype(bool)
        w = bit_width(self.type_id) // 8
        return raw[self.offset * w:(self.offset + self.length) * w].view(NP_OF[self.type_id]).copy()

    def validity_numpy(self):
        if self.validity is None:
            return np.ones(self.length, bool)
        raw = self.validity.cpu().numpy()
        return np.unpackbits(raw, bitorder="little")[self.offset:self.offset + self.length].astype(bool)
