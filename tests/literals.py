"""Replays tests/golden/reference_literals.json (the reference's own known-answer tests,
restated as data) through any backend with the orc.Api interface."""
import json
import os

import numpy as np

import orc
from orc import Arr, BOOL, NUMERIC, INTS, FLOAT, DOUBLE, DROP, EMIT_NULL, ASC, DESC

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_literals.json")
TYPE_GROUPS = {"numeric": NUMERIC, "integral": INTS, "real": [FLOAT, DOUBLE], "boolean": [BOOL],
               "int8": [orc.INT8], "uint8": [orc.UINT8], "float": [FLOAT], "double": [DOUBLE],
               "int64": [orc.INT64], "int32": [orc.INT32], "uint32": [orc.UINT32]}
CMP = {"equal": orc.EQ, "not_equal": orc.NE, "greater": orc.GT, "greater_equal": orc.GE, "less": orc.LT,
       "less_equal": orc.LE}
BOOL_OPS = {"invert": orc.INVERT, "and": orc.AND, "and_not": orc.AND_NOT, "or": orc.OR, "xor": orc.XOR,
            "and_kleene": orc.AND_KLEENE, "and_not_kleene": orc.AND_NOT_KLEENE, "or_kleene": orc.OR_KLEENE}
ARITH = {"add": orc.ADD, "subtract": orc.SUB, "multiply": orc.MUL, "divide": orc.DIV, "add_checked": orc.ADD_C,
         "subtract_checked": orc.SUB_C, "multiply_checked": orc.MUL_C, "divide_checked": orc.DIV_C}


def _num(x):
    if isinstance(x, str):
        return {"NaN": np.nan, "Inf": np.inf, "-Inf": -np.inf, "-0.0": -0.0}[x]
    return x


def arr(type_id, items, offset_pad=0):
    """JSON list (null = null slot) -> Arr; validity buffer only if a null is present
    (ArrayFromJSON builds one only then)."""
    vals = [0 if v is None else _num(v) for v in items]
    valid = [v is not None for v in items]
    data = np.array(vals, dtype=bool if type_id == BOOL else orc.NP_OF[type_id])
    if offset_pad:
        data = np.concatenate([np.zeros(offset_pad, data.dtype), data])
        valid = [True] * offset_pad + valid
    has_null = not all(valid)
    return Arr(type_id, data, np.array(valid, bool) if has_null else None, offset=offset_pad, length=len(items))


def logical(values, validity):
    """-> list with None at null slots (what AssertArraysEqual compares)"""
    out = []
    for v, ok in zip(values.tolist(), validity.tolist()):
        out.append(v if ok else None)
    return out


def expect(type_id, items):
    out = []
    for v in items:
        if v is None:
            out.append(None)
        elif type_id == BOOL:
            out.append(bool(v))
        else:
            out.append(orc.NP_OF[type_id](_num(v)).item())
    return out


def same_logical(a, b):
    if len(a) != len(b):
        return False
    for x, y in zip(a, b):
        if x is None or y is None:
            if x is not y:
                return False
        elif isinstance(x, float) and np.isnan(x):
            if not (isinstance(y, float) and np.isnan(y)):
                return False
        elif x != y:
            return False
    return True


def run_all(api, skip_grouper_multi_batch=False):
    """Runs every literal case; returns the list of failures (empty = pass)."""
    lit = json.load(open(PATH))
    fails = []

    def check(name, got, want):
        if not same_logical(got, want):
            fails.append("%s: got %s want %s" % (name, got, want))

    for grp in lit["filter"]:
        for t in TYPE_GROUPS[grp["types"]]:
            for c in grp["cases"]:
                v = arr(t, c["values"])
                f = arr(BOOL, c["filter"])
                if "filter_slice" in c:
                    o, n = c["filter_slice"]
                    f = Arr(BOOL, f.data, f.valid, o, n)
                for ns, key in ((EMIT_NULL, "emit_null"), (DROP, "drop")):
                    r = api.filter(v, f, ns)
                    check("%s t=%d %s %s" % (grp["source"], t, c, key), logical(r.values, r.validity() if r.has_validity else np.ones(r.length, bool)),
                          expect(t, c[key]))
    for c in lit["filter_errors"]:
        for ns in (DROP, EMIT_NULL):
            try:
                api.filter(arr(orc.INT32, c["values"]), arr(BOOL, c["filter"]), ns)
                fails.append("%s: no error" % c["source"])
            except orc.StatusError as e:
                if e.code != c["code"]:
                    fails.append("%s: code %d" % (c["source"], e.code))
    for grp in lit["take_indices"]:
        for c in grp["cases"]:
            ns = DROP if c["null_selection"] == "drop" else EMIT_NULL
            r = api.take_indices(arr(BOOL, c["filter"]), ns)
            if r.type_id != orc.UINT16:
                fails.append("%s: index type" % grp["source"])
            check("%s %s" % (grp["source"], c), logical(r.values, r.validity() if r.has_validity else np.ones(r.length, bool)),
                  expect(orc.UINT16, c["indices"]))
    for grp in lit["take"]:
        its = INTS if grp["index_types"] == "all" else TYPE_GROUPS[grp["index_types"]]
        for t in TYPE_GROUPS[grp["types"]]:
            for it in its:
                for c in grp["cases"]:
                    r = api.take(arr(t, c["values"]), arr(it, c["indices"]))
                    check("%s t=%d it=%d %s" % (grp["source"], t, it, c), logical(r.values, r.validity()), expect(t, c["expected"]))
    for c in lit["take_errors"]:
        for t in NUMERIC + [BOOL]:
            try:
                api.take(arr(t, [bool(x) for x in c["values"]] if t == BOOL else c["values"]), arr(orc.INT8, c["indices"]))
                fails.append("%s: no error" % c["source"])
            except orc.StatusError as e:
                if e.code != c["code"]:
                    fails.append("%s: code %d" % (c["source"], e.code))
    for grp in lit["sort_indices"]:
        for t in TYPE_GROUPS[grp["types"]]:
            for c in grp["cases"]:
                got = api.sort_indices(arr(t, c["values"]), ASC if c["order"] == "ascending" else DESC)
                if got.tolist() != c["indices"]:
                    fails.append("%s t=%d %s: got %s" % (grp["source"], t, c, got.tolist()))
    for c in lit["sort_indices_chunked"]:
        t = TYPE_GROUPS[c["type"]][0]
        chunks = [arr(t, ch) for ch in c["chunks"]]
        for order, key in ((ASC, "ascending"), (DESC, "descending")):
            got = api.sort_indices(chunks, order)
            if got.tolist() != c[key]:
                fails.append("%s %s: got %s" % (c["source"], key, got.tolist()))
    for c in lit["sort_indices_multi"]:
        ts = [TYPE_GROUPS[t][0] for t in c["types"]]
        keys = [[arr(t, ch) for ch in col] for t, col in zip(ts, c["columns"])]
        got = api.sort_indices_multi(keys, [ASC if o == "ascending" else DESC for o in c["orders"]])
        if got.tolist() != c["indices"]:
            fails.append("%s: got %s" % (c["source"], got.tolist()))
    for grp in lit["vector_hash"]:
        for t in TYPE_GROUPS[grp["types"]]:
            for c in grp["cases"]:
                a = arr(t, c["values"])
                if "slice" in c:
                    a = Arr(t, a.data, a.valid, c["slice"][0], c["slice"][1])
                if grp["fn"] == "unique":
                    u = api.unique(a)
                    check(grp["source"], logical(u.values, u.validity() if u.has_validity else np.ones(u.length, bool)), expect(t, c["unique"]))
                elif grp["fn"] == "value_counts":
                    u, n = api.value_counts(a)
                    check(grp["source"], logical(u.values, u.validity() if u.has_validity else np.ones(u.length, bool)), expect(t, c["unique"]))
                    check(grp["source"] + " counts", list(n.tolist()), c["counts"])
                else:
                    i, dic = api.dictionary_encode(a, 1)
                    check(grp["source"] + " dictionary", logical(dic.values, np.ones(dic.length, bool)), expect(t, c["dictionary"]))
                    check(grp["source"] + " indices", logical(i.values, i.validity() if i.has_validity else np.ones(i.length, bool)),
                          expect(orc.INT32, c["indices"]))
    for grp in lit["make_groupings"]:
        for c in grp["cases"]:
            off, idx = api.make_groupings(np.array(c["ids"], np.uint32), len(c["groupings"]))
            got = [idx[off[g]:off[g + 1]].tolist() for g in range(len(c["groupings"]))]
            if got != c["groupings"]:
                fails.append("%s %s: got %s" % (grp["source"], c["ids"], got))
    for grp in lit["grouper"]:
        for t in TYPE_GROUPS[grp["types"]]:
            batches = [[arr(t, b["keys"])] for b in grp["batches"]]
            ids, ng, uniq = api.group_ids(batches)
            for b, got in zip(grp["batches"], ids):
                if got.tolist() != b["ids"]:
                    fails.append("%s t=%d %s: got %s" % (grp["source"], t, b, got.tolist()))
    for c in lit["groupby"]:
        vt, kt = TYPE_GROUPS[c["argument_type"]][0], TYPE_GROUPS[c["key_type"]][0]
        v, k = arr(vt, c["argument"]), arr(kt, c["key"])
        ids, ng, uniq = api.group_ids([[k]])
        s = api.hash_aggregate("sum", v, ids[0], ng)
        check(c["source"] + " hash_sum", logical(s.values, s.validity()), expect(DOUBLE, c["hash_sum"]))
        n = api.hash_aggregate("count", v, ids[0], ng)
        check(c["source"] + " hash_count", n.values.tolist(), c["hash_count"])
        check(c["source"] + " key_0", logical(uniq[0][0], uniq[0][1]), expect(kt, c["key_0"]))
        s2 = api.hash_aggregate("sum", k, ids[0], ng)
        check(c["source"] + " hash_sum(key)", logical(s2.values, s2.validity()), expect(orc.INT64, c["hash_sum_of_key_as_argument"]))
        # the fused entry point on the same data (mask = all true)
        sums, counts, okeys, okv = api.filter_groupby_sum(v, k, Arr(BOOL, np.ones(len(c["key"]), bool)))
        check(c["source"] + " fused sums", [x if cnt > 0 else None for x, cnt in zip(sums.tolist(), counts.tolist())],
              expect(DOUBLE, c["hash_sum"]))
        check(c["source"] + " fused keys", logical(okeys, okv), expect(kt, c["key_0"]))
    for grp in lit["compare_array_scalar"]:
        for t in TYPE_GROUPS[grp["types"]]:
            for c in grp["cases"]:
                if not c["lhs"]:
                    continue
                r = api.compare(CMP[c["op"]], arr(t, c["lhs"]), grp["scalar"])
                check("%s t=%d %s" % (grp["source"], t, c), logical(r.values, r.validity()),
                      [None if x is None else bool(x) for x in c["expected"]])
    for grp in lit["arithmetic"]:
        for t in TYPE_GROUPS[grp["types"]]:
            for opname in grp["ops"]:
                for c in grp["cases"]:
                    lhs = arr(t, c["lhs"]) if "lhs" in c else c["lhs_scalar"]
                    rhs = arr(t, c["rhs"]) if "rhs" in c else c["rhs_scalar"]
                    if isinstance(lhs, Arr) and lhs.length == 0:
                        continue
                    r = api.arith(ARITH[opname], lhs, rhs)
                    check("%s t=%d %s %s" % (grp["source"], t, opname, c), logical(r.values, r.validity()), expect(t, c["expected"]))
    # ---- mask-building scalars ----
    def lv(r):
        return logical(r.values, r.validity() if r.has_validity else np.ones(r.length, bool))

    for c in lit.get("boolean", []):
        op = BOOL_OPS[c["op"]]
        left = arr(BOOL, c["left"])
        r = api.boolean(op, left) if op == orc.INVERT else api.boolean(op, left, arr(BOOL, c["right"]))
        check("%s" % c["source"], lv(r), expect(BOOL, c["expected"]))
        if c["op"] in lit["boolean_scalar_equivalence"]["ops"]:
            n = len(c["left"])
            for sc in (None, True, False):  # CheckBooleanScalarArrayBinary :37-55
                const = Arr(BOOL, np.full(n, bool(sc)), None if sc is not None else np.zeros(n, bool))
                check("%s scalar(%s), array" % (c["source"], sc), lv(api.boolean(op, sc, left)), lv(api.boolean(op, const, left)))
                check("%s array, scalar(%s)" % (c["source"], sc), lv(api.boolean(op, left, sc)), lv(api.boolean(op, left, const)))
    for grp in lit.get("validity", []):
        for t in TYPE_GROUPS[grp["types"]]:
            for c in grp["cases"]:
                f = api.is_valid if grp["func"] == "is_valid" else api.is_null
                r = f(arr(t, c["in"]))
                if r.has_validity or r.null_count != 0:
                    fails.append("%s: the result must not carry nulls" % grp["source"])
                check("%s %s" % (grp["source"], c), lv(r), expect(BOOL, c["out"]))
    if "is_nan" in lit:
        grp = lit["is_nan"]
        for t in TYPE_GROUPS[grp["types"]]:
            for c in grp["cases"]:
                check("%s t=%d %s" % (grp["source"], t, c), lv(api.is_nan(arr(t, c["in"]))), expect(BOOL, c["out"]))
    for grp in lit.get("fill_null", []):
        for t in TYPE_GROUPS[grp["types"]]:
            for c in grp["cases"]:
                r = api.fill_null(arr(t, c["in"]), grp["fill"])
                check("%s t=%d %s" % (grp["source"], t, c), lv(r), expect(t, c["out"]))
                if grp["fill"] is not None and r.length and (r.has_validity or r.null_count != 0):
                    fails.append("%s: null_count must be 0 (FillNullSetsZeroNullCount :175-183)" % grp["source"])
    # ---- scalar aggregates; the Finalize rules (aggregate_basic_internal.h:73-97, 254-267)
    # are applied here exactly as the host layer applies them on top of one amdarrow_aggregate call ----
    for grp in lit.get("aggregate", []):
        for t in TYPE_GROUPS[grp["types"]]:
            for c in grp["cases"]:
                a = arr(t, c["in"])
                r = api.aggregate(a)
                n, vc = a.length, r["valid_count"]
                name = "%s t=%d %s" % (grp["source"], t, c)
                if "sum" in c:
                    got = None if vc == 0 else r["sum"]
                    if got != (None if c["sum"] is None else type(got)(c["sum"]) if got is not None else c["sum"]):
                        fails.append("%s: sum %s" % (name, got))
                if "mean" in c:
                    got = None if vc == 0 else float(r["sum"]) / vc
                    if got != c["mean"]:
                        fails.append("%s: mean %s" % (name, got))
                if "count" in c and [vc, n - vc] != c["count"]:
                    fails.append("%s: count %s" % (name, [vc, n - vc]))
                for key, emit_null in (("min_max", False), ("min_max_emit_null", True)):
                    if key in c:
                        got = None if vc == 0 or (emit_null and vc < n) else [r["min"], r["max"]]
                        want = None if c[key] is None else [_num(x) if not isinstance(x, bool) else int(x) for x in c[key]]
                        if got != want:
                            fails.append("%s: %s %s" % (name, key, got))
    return fails
