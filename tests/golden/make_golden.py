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


def nested_cases():
    """The seeded case list of tests/golden/ref_nested.*: (values tree, selection, is_filter, option)."""
    import nested as T
    rng = np.random.default_rng(0x0ff1ce + 3)
    kinds = [T.K_BINARY, T.K_UTF8, T.K_LARGE_BINARY, T.K_LARGE_UTF8, T.K_FSB, T.K_DEC128, T.K_DEC256, T.K_LIST, T.K_LARGE_LIST, T.K_MAP,
             T.K_FSL, T.K_STRUCT, T.K_DICT, T.K_NULL]
    for kind in kinds:
        for trial in range(10):
            n = int(rng.integers(1, 200)) if trial else 0
            if kind == T.K_DICT:
                n = max(n, 1)  # (an empty dictionary array comes back from the executor with an empty dictionary)
            off = int(rng.choice([0, 0, 3])) if kind != T.K_FSL else 0  # FSLImpl ignores the array's offset (:1704): pinned at 0
            v = T.rand_node(rng, kind, n, float(rng.choice([0, 0.1, 0.5])), off)
            m = orc.random_arr(rng, orc.BOOL, n, float(rng.choice([0, 0.2])), int(rng.choice([0, 5])))
            yield v, m, 1, int(rng.integers(0, 2))
            if n:
                yield v, T.rand_indices(rng, n, int(rng.integers(0, 150)), float(rng.choice([0, 0.2])), int(rng.choice([0, 2]))), 0, 1
            else:
                yield v, orc.Arr(orc.INT32, np.zeros(3, np.int32), np.zeros(3, bool)), 0, 1
    # bounds errors (TakeExec checks before anything else, :1330-1345)
    v = T.rand_node(rng, T.K_LIST, 10, 0.1)
    yield v, orc.Arr(orc.INT64, np.array([1, 2, 10, 3], np.int64)), 0, 1
    yield v, orc.Arr(orc.INT8, np.array([1, -2, 4], np.int8)), 0, 1
    yield T.rand_node(rng, T.K_NULL, 5), orc.Arr(orc.INT32, np.array([0, 7], np.int32)), 0, 1


def tree_to_json(a, put):
    return {"kind": a.kind, "length": a.length, "offset": a.offset, "null_count": a.null_count, "type_id": a.type_id,
            "byte_width": a.byte_width, "list_size": a.list_size, "valid": None if a.valid is None else put(a.valid),
            "buf1": None if a.buf1 is None else put(a.buf1), "buf2": None if a.buf2 is None else put(a.buf2),
            "children": [tree_to_json(c, put) for c in a.children]}


def main_nested():
    """tests/golden/ref_nested.{npz,json}: take / filter of non-primitive arrays through the real reference."""
    import nested as T
    ref = T.RefNested()
    blob, manifest = bytearray(), []

    def put(arr):
        arr = np.asarray(arr)
        is_bool = arr.dtype == bool
        raw = np.packbits(arr, bitorder="little") if is_bool else np.ascontiguousarray(arr)
        off = len(blob)
        blob.extend(raw.tobytes())
        return [off, "bool" if is_bool else raw.dtype.str, int(arr.size)]

    for v, sel, is_filter, option in nested_cases():
        entry = {"values": tree_to_json(v, put), "is_filter": is_filter, "option": option,
                 "selection": {"type": sel.type_id, "offset": sel.offset, "length": sel.length, "null_count": sel.null_count,
                               "data": put(sel.data), "valid": None if sel.valid is None else put(sel.valid)}}
        try:
            entry["result"] = tree_to_json(ref.select(v, sel, is_filter, option), put)
        except orc.StatusError as e:
            entry["error"] = [e.code, e.msg]
        manifest.append(entry)
    groups = []
    for batches in T.group_cases():  # Grouper over variable-width / dictionary key columns (hash_aggregate.cc:198-311)
        ids, ng, uniq = ref.group(batches)
        groups.append({"batches": [[tree_to_json(c, put) for c in cols] for cols in batches], "ids": [put(i) for i in ids],
                       "num_groups": int(ng), "uniques": [tree_to_json(u, put) for u in uniq]})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_nested")
    np.savez_compressed(out + ".npz", blob=np.frombuffer(bytes(blob), np.uint8))
    with open(out + ".json", "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (main_nested)", "cases": manifest, "group_cases": groups}, f, separators=(",", ":"))
    print("nested: wrote %d cases, blob %.1f KiB" % (len(manifest), os.path.getsize(out + ".npz") / 1024))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "nested":
        main_nested()
    else:
        main()
        main_nested()
