"""arrow-1_amd -- MI355X (gfx950) backend for the Arrow compute hot path.

Layout:  csrc/ (hand-written HIP kernels + the C ABI of include/amdarrow.h),
         host/ (C++ mirror of arrow::compute: CallFunction / FunctionRegistry / ExecContext),
         compute.py / device.py / dist.py (Python mirror + torch.distributed sharding).
The directory name is not a Python identifier; load it with `__graft_entry__.load_package()`
(it registers the package as `arrow1_amd`)."""
from . import _abi
from ._abi import *  # noqa: F401,F403  (type ids / option enums)
from ._abi import AmdArrowError, load_library

__all__ = ["_abi", "AmdArrowError", "load_library"]
