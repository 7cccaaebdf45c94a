"""ctypes declarations for include/amdarrow.h.  The library is REQUIRED: importing the
package without a built arrow-1_amd/libamdarrow.so raises -- there is no CPU fallback."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libamdarrow.so")

POWER, POWER_CHECKED, NEGATE, NEGATE_CHECKED = 8, 9, 10, 11  # after add ... divide_checked (0-7)
BOOL, UINT8, INT8, UINT16, INT16, UINT32, INT32, UINT64, INT64, FLOAT, DOUBLE = 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12
DROP, EMIT_NULL = 0, 1
ASCENDING, DESCENDING = 0, 1
COUNT_NON_NULL, COUNT_NULL = 0, 1
MINMAX_SKIP, MINMAX_EMIT_NULL = 0, 1
DICT_ENCODE, DICT_MASK = 0, 1  # DictionaryEncodeOptions::NullEncodingBehavior, api_vector.h:67-82
EQ, NE, GT, GE, LT, LE = range(6)
ADD, SUB, MUL, DIV, ADD_CHECKED, SUB_CHECKED, MUL_CHECKED, DIV_CHECKED = range(8)
HASH_SUM, HASH_COUNT, HASH_MEAN = 0, 1, 2
INVERT, AND, AND_NOT, OR, XOR, AND_KLEENE, AND_NOT_KLEENE, OR_KLEENE = range(8)
SCALAR_NONE, SCALAR_FALSE, SCALAR_TRUE, SCALAR_NULL = -1, 0, 1, 2

STATUS_NAMES = {0: "OK", 1: "OutOfMemory", 2: "KeyError", 3: "TypeError", 4: "Invalid", 6: "CapacityError",
                7: "IndexError", 9: "UnknownError", 10: "NotImplemented"}


class CArray(C.Structure):
    _fields_ = [("values", C.c_void_p), ("validity", C.c_void_p), ("length", C.c_int64),
                ("offset", C.c_int64), ("type_id", C.c_int32), ("null_count", C.c_int64)]


class CBinaryArray(C.Structure):
    _fields_ = [("offsets", C.c_void_p), ("data", C.c_void_p), ("validity", C.c_void_p), ("length", C.c_int64),
                ("offset", C.c_int64), ("null_count", C.c_int64)]


class CVarArray(C.Structure):
    _fields_ = [("offsets", C.c_void_p), ("offset_bits", C.c_int32), ("data", C.c_void_p), ("validity", C.c_void_p),
                ("length", C.c_int64), ("offset", C.c_int64), ("null_count", C.c_int64)]


class COut(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("has_validity", C.c_int32)]


class CAgg(C.Structure):
    _fields_ = [("valid_count", C.c_int64), ("sum", C.c_uint64), ("min", C.c_uint64), ("max", C.c_uint64)]


# every symbol include/amdarrow.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_A = C.POINTER(CArray)
_O = C.POINTER(COut)
SYMBOLS = {
    "amdarrow_abi_version": (C.c_int, []),
    "amdarrow_last_error": (C.c_char_p, []),
    "amdarrow_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "amdarrow_ctx_create": (C.c_int, [C.c_int, _P, C.c_int, C.POINTER(_P)]),
    "amdarrow_ctx_destroy": (C.c_int, [_P]),
    "amdarrow_ctx_synchronize": (C.c_int, [_P]),
    "amdarrow_ctx_stream": (_P, [_P]),
    "amdarrow_malloc": (C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    "amdarrow_free": (C.c_int, [_P, _P]),
    "amdarrow_bytes_allocated": (C.c_int64, [_P]),
    "amdarrow_max_memory": (C.c_int64, [_P]),
    "amdarrow_pool_cached_bytes": (C.c_int64, [_P]),
    "amdarrow_pool_trim": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "amdarrow_pool_set_cache_limit": (C.c_int, [_P, C.c_int64]),
    "amdarrow_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_int64]),
    "amdarrow_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_int64]),
    "amdarrow_memcpy_d2d": (C.c_int, [_P, _P, _P, C.c_int64]),
    "amdarrow_memset": (C.c_int, [_P, _P, C.c_int, C.c_int64]),
    "amdarrow_count_set_bits": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    "amdarrow_filter_output_size": (C.c_int, [_P, _A, C.c_int, C.POINTER(C.c_int64)]),
    "amdarrow_filter": (C.c_int, [_P, _A, _A, C.c_int, _P, _P, C.c_int64, _O]),
    "amdarrow_take_indices": (C.c_int, [_P, _A, C.c_int, _P, _P, C.c_int64, C.POINTER(C.c_int), _O]),
    "amdarrow_take": (C.c_int, [_P, _A, _A, C.c_int, _P, _P, _O]),
    "amdarrow_sort_indices": (C.c_int, [_P, _A, C.c_int, C.c_int, _P]),
    "amdarrow_binary_concatenate_size": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "amdarrow_binary_concatenate": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _O]),
    "amdarrow_binary_take_offsets": (C.c_int, [_P, C.POINTER(CBinaryArray), _A, C.c_int, _P, _P, C.POINTER(C.c_int64), _O]),
    "amdarrow_binary_take_data": (C.c_int, [_P, C.POINTER(CBinaryArray), _A, _P, _P]),
    "amdarrow_sort_indices_multi": (C.c_int, [_P, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, _P]),
    "amdarrow_concatenate": (C.c_int, [_P, _P, C.c_int, _P, _P, _O]),
    "amdarrow_compare": (C.c_int, [_P, C.c_int, _A, _P, _A, _P, _P, _P, _O]),
    "amdarrow_arith": (C.c_int, [_P, C.c_int, _A, _P, _A, _P, _P, _P, _O]),
    "amdarrow_boolean": (C.c_int, [_P, C.c_int, _A, C.c_int, _A, C.c_int, _P, _P, _O]),
    "amdarrow_validity_test": (C.c_int, [_P, C.c_int, _A, _P, _O]),
    "amdarrow_is_nan": (C.c_int, [_P, _A, _P, _P, _O]),
    "amdarrow_fill_null": (C.c_int, [_P, _A, _P, _P, _P, _O]),
    "amdarrow_cast_numeric": (C.c_int, [_P, _A, C.c_int, C.c_int, _P]),
    "amdarrow_aggregate": (C.c_int, [_P, _A, _P]),
    "amdarrow_grouper_create": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(_P)]),
    "amdarrow_grouper_destroy": (C.c_int, [_P]),
    "amdarrow_grouper_consume": (C.c_int, [_P, _A, _P]),
    "amdarrow_grouper_lookup": (C.c_int, [_P, _A, _P]),
    "amdarrow_grouper_num_groups": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "amdarrow_grouper_uniques": (C.c_int, [_P, C.c_int, _P, _P, _O]),
    "amdarrow_make_groupings": (C.c_int, [_P, _P, C.c_int64, C.c_uint32, _P, _P]),
    "amdarrow_hash_aggregate": (C.c_int, [_P, C.c_int, _A, _P, C.c_int64, C.c_int, _P, _P, _O]),
    "amdarrow_dense_merge_scatter": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, _P, _P]),
    "amdarrow_dense_merge_order": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int64,
                                             _P, _P, _P, _P, _P]),
    "amdarrow_dictionary_indices": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, _P, _P, _O]),
    "amdarrow_hash_minmax_init": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, C.c_int64]),
    "amdarrow_hash_minmax_accumulate": (C.c_int, [_P, _A, _P, _P, _P, _P, _P]),
    "amdarrow_hash_minmax_finalize": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P, _P, _O]),
    "amdarrow_hash_accumulate": (C.c_int, [_P, C.c_int, _A, _P, C.c_int, _P, _P]),
    "amdarrow_hash_finalize": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, _P, _P, _P, _O]),
    "amdarrow_filter_groupby_multi": (C.c_int, [_P, _A, _P, C.c_int, _A, C.c_int64, _P, _P, _P, _P, C.POINTER(C.c_int64), _P,
                                                C.POINTER(C.c_int64)]),
    "amdarrow_filter_groupby": (C.c_int, [_P, _A, _A, _A, C.c_int64, _P, _P, _P, _P, _P, C.POINTER(C.c_int64)]),
    "amdarrow_groupby_last_path": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "amdarrow_take_rows": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int64, _A, C.c_int, _P, _P, _O]),
    "amdarrow_gather_fixed_width": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, C.c_int64, _P]),
    "amdarrow_var_take_offsets": (C.c_int, [_P, C.POINTER(CVarArray), _P, C.c_int64, _P, C.POINTER(C.c_int64)]),
    "amdarrow_var_take_data": (C.c_int, [_P, C.POINTER(CVarArray), _P, C.c_int64, _P, _P]),
    "amdarrow_list_take_child_indices": (C.c_int, [_P, C.POINTER(CVarArray), _P, C.c_int64, _P, _P]),
    "amdarrow_binary_grouper_create": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "amdarrow_binary_grouper_destroy": (C.c_int, [_P]),
    "amdarrow_binary_grouper_consume": (C.c_int, [_P, C.POINTER(CVarArray), _P]),
    "amdarrow_binary_grouper_num_groups": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "amdarrow_binary_grouper_uniques_size": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "amdarrow_binary_grouper_uniques": (C.c_int, [_P, _P, _P, _P, _O]),
    "amdarrow_fixed_size_list_child_indices": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int64, _P, _P]),
}


class AmdArrowError(RuntimeError):
    """A non-OK arrow::StatusCode returned by the C ABI."""

    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS_NAMES.get(code, str(code)), msg))
        self.code, self.msg = code, msg


_lib = None


def load_library(path=None):
    """dlopen libamdarrow.so and bind every declared symbol.  Raises if it is missing:
    the HIP library IS the product path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or make -C arrow-1_amd/csrc).  There is no CPU fallback." % p)
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.amdarrow_abi_version() != 1:
        raise ImportError("amdarrow ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


def check(lib, rc):
    if rc != 0:
        raise AmdArrowError(rc, lib.amdarrow_last_error().decode())
