#!/usr/bin/env python3
"""Generates tests/golden/ref_random.{npz,json}: inputs + outputs of the REAL reference
(oracle/_ref, built from /root/reference by oracle/ref_build.mk) for the case list in
tests/golden_io.py.  Run in the build container only:  python tests/golden/make_golden.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import golden_io  # noqa: E402
import orc  # noqa: E402


def main():
    ref = orc.Api("ref")
    blob, manifest = bytearray(), []

    def put(arr):
        """append one array to the blob -> [offset, dtype, count, is_bool_bits]"""
        arr = np.asarray(arr)
        is_bool = arr.dtype == bool
        raw = np.packbits(arr, bitorder="little") if is_bool else np.ascontiguousarray(arr)
        off = len(blob)
        blob.extend(raw.tobytes())
        return [off, "bool" if is_bool else raw.dtype.str, int(arr.size)]

    for op, params, arrs in golden_io.iter_cases():
        entry = {"op": op, "params": params, "inputs": [], "outputs": {}}
        for a in arrs:
            if isinstance(a, orc.BinArr):
                entry["inputs"].append({"type": "binary", "offset": a.offset, "length": a.length, "null_count": a.null_count,
                                        "offsets": put(a.offsets), "data": put(a.data),
                                        "valid": None if a.valid is None else put(a.valid)})
                continue
            entry["inputs"].append({"type": a.type_id, "offset": a.offset, "length": a.length,
                                    "null_count": a.null_count, "data": put(a.data),
                                    "valid": None if a.valid is None else put(a.valid)})
        for k, v in golden_io.run_case(ref, op, params, arrs).items():
            entry["outputs"][k] = put(v)
        manifest.append(entry)
    np.savez_compressed(golden_io.NPZ, blob=np.frombuffer(bytes(blob), np.uint8))
    with open(golden_io.MANIFEST, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "reference": "zhztheplayer/arrow-1 (Arrow 5.0.0-SNAPSHOT) compute kernels built by oracle/ref_build.mk",
                   "cases": manifest}, f, separators=(",", ":"))
    print("wrote %d cases, blob %.1f KiB, manifest %.1f KiB" % (len(manifest), os.path.getsize(golden_io.NPZ) / 1024,
                                                              os.path.getsize(golden_io.MANIFEST) / 1024))


if __name__ == "__main__":
    main()
