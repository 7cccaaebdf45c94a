"""CPU: the oracle restatement against (1) the reference's own known-answer vectors,
(2) committed outputs of the real reference on seeded random inputs, (3) the live reference
where oracle/_ref is present."""
import numpy as np
import pytest

import golden_io
import literals
import orc


def test_oracle_matches_reference_literals(oracle):
    fails = literals.run_all(oracle)
    assert not fails, "\n".join(fails[:20])


def test_reference_matches_its_own_literals(reference):
    """Sanity of the harness itself: the real reference passes its own vectors through our driver."""
    fails = literals.run_all(reference)
    assert not fails, "\n".join(fails[:20])


def test_oracle_matches_committed_reference_outputs(oracle):
    cases = golden_io.load_golden()
    assert len(cases) > 1000
    bad = []
    for i, (op, params, arrs, want) in enumerate(cases):
        got = golden_io.run_case(oracle, op, params, arrs)
        if "error" in want or "error" in got:
            ok = "error" in want and "error" in got and np.array_equal(got["error"], want["error"]) and \
                np.array_equal(got["msg"], want["msg"])
            msg = "error differs"
        else:
            ok, msg = golden_io.same_output(got, want)
        if not ok:
            bad.append("case %d %s %s: %s" % (i, op, params, msg))
    assert not bad, "\n".join(bad[:20])


def test_committed_fixtures_are_reproducible(reference):
    """Regenerating a sample of the fixtures from the live reference gives the committed bytes."""
    cases = golden_io.load_golden()
    for (op, params, arrs, want) in cases[::37]:
        got = golden_io.run_case(reference, op, params, arrs)
        ok, msg = golden_io.same_output(got, want) if "error" not in want else (True, "")
        assert ok, (op, params, msg)


def test_oracle_matches_live_reference_fresh_seed(oracle, reference):
    rng = np.random.default_rng(20260101)
    R = orc.random_arr
    for t in [orc.BOOL, orc.INT16, orc.INT64, orc.FLOAT]:
        for ns in (orc.DROP, orc.EMIT_NULL):
            v, f = R(rng, t, 777, 0.1, 5), R(rng, orc.BOOL, 777, 0.1, 9)
            assert oracle.filter(v, f, ns).same(reference.filter(v, f, ns))
            assert oracle.take_indices(f, ns).same(reference.take_indices(f, ns))
        i = R(rng, orc.INT32, 300, 0.1, 1, lo=0, hi=776)
        tv = R(rng, t, 777, 0.1, 5)
        assert oracle.take(tv, i).same(reference.take(tv, i))
        if t != orc.BOOL:
            a = R(rng, t, 999, 0.1, 2, lo=-9, hi=9)
            for order in (orc.ASC, orc.DESC):
                assert np.array_equal(oracle.sort_indices(a, order), reference.sort_indices(a, order))
    v = R(rng, orc.INT64, 4000, 0.02)
    k = R(rng, orc.INT32, 4000, 0.02, lo=0, hi=99)
    m = R(rng, orc.BOOL, 4000, 0.0)
    for x, y in zip(oracle.filter_groupby_sum(v, k, m), reference.filter_groupby_sum(v, k, m)):
        assert np.array_equal(x, y)


def test_edge_cases(oracle):
    e = orc.Arr(orc.INT32, np.zeros(0, np.int32))
    f = orc.Arr(orc.BOOL, np.zeros(0, bool))
    assert oracle.filter(e, f).length == 0
    assert oracle.sort_indices(e).tolist() == []
    assert oracle.take(orc.Arr(orc.INT32, [1, 2, 3]), orc.Arr(orc.UINT8, np.zeros(0, np.uint8))).length == 0
    with pytest.raises(orc.StatusError) as ei:
        oracle.take(orc.Arr(orc.INT32, [1, 2, 3]), orc.Arr(orc.INT64, [3]))
    assert ei.value.code == orc.INDEX_ERROR and "Index 3 out of bounds" in ei.value.msg
    with pytest.raises(orc.StatusError) as ei:
        oracle.arith(orc.DIV, orc.Arr(orc.INT32, [1, 2]), orc.Arr(orc.INT32, [1, 0]))
    assert ei.value.code == orc.INVALID and ei.value.msg == "divide by zero"
    with pytest.raises(orc.StatusError) as ei:
        oracle.arith(orc.ADD_C, orc.Arr(orc.INT8, [127]), orc.Arr(orc.INT8, [1]))
    assert ei.value.msg == "overflow"
    # unchecked add wraps (SafeSignedAdd), int MIN / -1 unchecked -> 0
    assert oracle.arith(orc.ADD, orc.Arr(orc.INT8, [127]), orc.Arr(orc.INT8, [1])).values.tolist() == [-128]
    assert oracle.arith(orc.DIV, orc.Arr(orc.INT8, [-128]), orc.Arr(orc.INT8, [-1])).values.tolist() == [0]


def test_mask_scalars_match_live_reference(oracle, reference):
    """SURVEY 8f.1 kernels (boolean / is_valid / is_null / is_nan / fill_null) on a fresh seed,
    every operand shape incl. null scalars, unknown null counts and empty inputs.  Kleene
    kernels with nulls are checked where the reference itself is sound: bit offsets that are
    multiples of 64 or at most 128 rows (see golden_io.iter_cases)."""
    rng = np.random.default_rng(20260102)
    R = orc.random_arr
    for n, lo, ro in [(0, 0, 0), (1, 2, 1), (77, 3, 9), (128, 5, 0), (500, 0, 0), (500, 64, 192)]:
        for p1, p2 in [(0, 0), (.3, 0), (0, .3), (.3, .3), (1.0, .3)]:
            for nc in (None, -1):
                a = R(rng, orc.BOOL, n, p1, lo, null_count=nc, force_validity=(nc == -1))
                b = R(rng, orc.BOOL, n, p2, ro, null_count=nc, force_validity=(nc == -1))
                for op in range(8):
                    assert oracle.boolean(op, a, b).same(reference.boolean(op, a, b)), (op, n, p1, p2)
                    for sc in (False, True, None):
                        assert oracle.boolean(op, a, sc).same(reference.boolean(op, a, sc)), (op, sc)
                        if op != orc.INVERT:
                            assert oracle.boolean(op, sc, b).same(reference.boolean(op, sc, b)), (op, sc)
        for t in [orc.BOOL, orc.INT8, orc.UINT16, orc.INT32, orc.INT64, orc.FLOAT, orc.DOUBLE]:
            for p in (0, .2, 1.0):
                v = R(rng, t, n, p, lo)
                assert oracle.is_valid(v).same(reference.is_valid(v))
                assert oracle.is_null(v).same(reference.is_null(v))
                for fill in (None, 1, 0):
                    assert oracle.fill_null(v, fill).same(reference.fill_null(v, fill)), (t, p, fill)
                if t in (orc.FLOAT, orc.DOUBLE):
                    d = v.data.copy()
                    d[::5] = np.nan
                    w = orc.Arr(t, d, v.valid, lo, n)
                    assert oracle.is_nan(w).same(reference.is_nan(w))


def test_scalar_aggregates_match_live_reference(oracle, reference):
    """SURVEY 8f.2 on a fresh seed: the restatement reproduces the reference bit for bit,
    including its pairwise float summation (aggregate_internal.h:66-139)."""
    rng = np.random.default_rng(20260103)
    R = orc.random_arr
    for t in [orc.BOOL] + orc.NUMERIC:
        for n, p, off in [(0, 0, 0), (1, 0, 0), (17, .3, 2), (1000, 0, 0), (4099, .05, 5), (4099, 1.0, 0)]:
            a = R(rng, t, n, p, off)
            x, y = reference.aggregate(a), oracle.aggregate(a)
            assert all(x[k] == y[k] or (x[k] != x[k] and y[k] != y[k]) for k in x), (t, n, p, x, y)


def test_multi_key_sort_and_concatenate_match_live_reference(oracle, reference):
    """SURVEY 8f.3 / a5 on a fresh seed.  The same keys go through all three reference sorters:
    RadixRecordBatchSorter (batch, <= 8 keys), MultipleKeyRecordBatchSorter (batch, > 8 keys) and
    MultipleKeyTableSorter (table) -- one order (MultipleKeyComparator, vector_sort.cc:1049-1180)."""
    import ctypes as C
    rng = np.random.default_rng(20260104)
    R = orc.random_arr
    types = [orc.INT8, orc.UINT8, orc.INT32, orc.UINT32, orc.INT64, orc.UINT64, orc.FLOAT, orc.DOUBLE, orc.INT16, orc.UINT16]

    def col(t, n, p):
        a = R(rng, t, n, p, int(rng.integers(0, 9)), lo=0, hi=3)
        if t in (orc.FLOAT, orc.DOUBLE) and n:
            d = a.data.copy()
            d[rng.random(len(d)) < 0.15] = np.nan
            d[rng.random(len(d)) < 0.15] = -0.0
            a = orc.Arr(t, d, a.valid, a.offset, n)
        return a
    for n in (0, 1, 50, 1500):
        for nkeys in (1, 2, 4, 10):
            for p in (0.0, 0.2):
                keys = [col(types[int(rng.integers(0, len(types)))], n, p) for _ in range(nkeys)]
                orders = [int(rng.integers(0, 2)) for _ in range(nkeys)]
                want = reference.sort_indices_multi(keys, orders)
                assert np.array_equal(oracle.sort_indices_multi(keys, orders), want), (n, nkeys, p)
                # the table sorter on the same single-chunk columns
                flat = (orc.CArr * nkeys)(*[k.c() for k in keys])
                nch = (C.c_int * nkeys)(*[1] * nkeys)
                ords = (C.c_int * nkeys)(*orders)
                out = np.zeros(n + 1, np.uint64)
                assert reference.lib.ref_sort_indices_multi_table(flat, nch, ords, nkeys, out.ctypes.data_as(C.c_void_p)) == 0
                assert np.array_equal(out[:n], want), ("table", n, nkeys, p)
    for t in [orc.BOOL, orc.INT16, orc.FLOAT, orc.INT64]:
        for nc in (None, -1):
            chunks = [R(rng, t, m, p, off, null_count=nc, force_validity=(nc == -1))
                      for m, p, off in [(5, 0, 0), (0, 0, 1), (200, .2, 3), (64, 0, 64), (77, 1.0, 9)]]
            assert oracle.concatenate(chunks).same(reference.concatenate(chunks)), (t, nc)
            assert oracle.concatenate(chunks[:2]).same(reference.concatenate(chunks[:2])), (t, nc)
    with pytest.raises(orc.StatusError):
        oracle.sort_indices_multi([], [])


def test_hash_min_max_matches_live_reference(oracle, reference):
    """SURVEY 8f.2 (rest) on a fresh seed, both null handlings, NaN / -0.0 / empty groups: the
    restatement is bit-identical (same sequential std::min / std::max as :667-668)."""
    rng = np.random.default_rng(20260105)
    R = orc.random_arr
    for t in orc.NUMERIC:
        for n, ng, p in [(0, 0, 0), (0, 5, 0), (1, 1, 0), (2000, 64, .2), (2000, 9, 1.0)]:
            v = R(rng, t, n, p, 5)
            if t in (orc.FLOAT, orc.DOUBLE) and n > 10:
                d = v.data.copy()
                d[rng.random(n + 5) < 0.1] = np.nan
                d[rng.random(n + 5) < 0.1] = -0.0
                d[rng.random(n + 5) < 0.1] = 0.0
                v = orc.Arr(t, d, v.valid, 5, n)
            ids = rng.integers(0, max(ng - 1, 1), n).astype(np.uint32)
            for mode in (0, 1):
                for a, b in zip(oracle.hash_min_max(v, ids, ng, mode), reference.hash_min_max(v, ids, ng, mode)):
                    assert a.same(b), (t, n, ng, p, mode)


def test_vector_hash_functions_match_live_reference(oracle, reference):
    """SURVEY 8f.4 on a fresh seed: unique / value_counts / dictionary_encode (both null encodings).
    Values under a null slot are unspecified in the reference (the memo table's null entry is not
    initialised for 16-bit and wider types), so they are compared where valid."""
    rng = np.random.default_rng(20260106)
    R = orc.random_arr
    for t in [orc.BOOL] + orc.NUMERIC:
        for n, p, off in [(0, 0, 0), (1, 0, 0), (1, 1.0, 0), (300, 0, 3), (300, .2, 5), (300, 1.0, 0), (5000, .05, 1)]:
            a = R(rng, t, n, p, off, lo=0, hi=30)
            if t in (orc.FLOAT, orc.DOUBLE) and n > 10:
                d = a.data.copy()
                d[::7], d[::11], d[::13] = np.nan, -0.0, 0.0
                a = orc.Arr(t, d, a.valid, off, n)
            assert oracle.unique(a).same_valid(reference.unique(a)), (t, n, p)
            (u1, c1), (u2, c2) = oracle.value_counts(a), reference.value_counts(a)
            assert u1.same_valid(u2) and np.array_equal(c1, c2), (t, n, p)
            for ne in (0, 1):
                (i1, d1), (i2, d2) = oracle.dictionary_encode(a, ne), reference.dictionary_encode(a, ne)
                assert i1.same_valid(i2) and d1.same_valid(d2), (t, n, p, ne)


def _rand_bin(rng, n, p_null, off=0, maxlen=12):
    items = []
    for _ in range(n + off):
        items.append(None if rng.random() < p_null else bytes(rng.integers(97, 123, int(rng.integers(0, maxlen + 1))).astype(np.uint8)))
    return orc.BinArr(items, off, n, pad=int(rng.integers(0, 5)))


def test_binary_take_and_filter_match_live_reference(oracle, reference):
    """SURVEY 8f.4 on a fresh seed: Take / Filter of BinaryArrays (32-bit offsets) incl. slices, null values,
    null indices / filter slots under both null selections, every index width, and the IndexError text."""
    rng = np.random.default_rng(20260107)
    for n in (0, 1, 7, 200, 3000):
        for pv in (0, .2, 1.0):
            v = _rand_bin(rng, n, pv, off=int(rng.integers(0, 4)))
            for pf in (0, .15):
                f = orc.random_arr(rng, orc.BOOL, n, pf, int(rng.integers(0, 9)))
                for ns in (orc.DROP, orc.EMIT_NULL):
                    assert oracle.binary_filter(v, f, ns).same(reference.binary_filter(v, f, ns)), (n, pv, pf, ns)
            if n:
                for it in orc.INTS:
                    i = orc.random_arr(rng, it, int(rng.integers(0, 300)), .1, 2, lo=0, hi=min(n - 1, 100))
                    assert oracle.binary_take(v, i).same(reference.binary_take(v, i)), (n, pv, it)
    for api in (oracle, reference):
        with pytest.raises(orc.StatusError) as ei:
            api.binary_take(orc.BinArr([b"a", b"bb"]), orc.Arr(orc.INT32, [0, 2]))
        assert ei.value.code == orc.INDEX_ERROR and ei.value.msg == "Index 2 out of bounds"


def test_binary_concatenate_matches_live_reference(oracle, reference):
    rng = np.random.default_rng(20260108)
    for sizes in ([(5, 0, 0), (0, 0, 1), (200, .2, 3), (64, 0, 2), (77, 1.0, 1)], [(3, 0, 0)], [(0, 0, 0)], [(10, .5, 2), (10, 0, 0)]):
        chunks = [_rand_bin(rng, n, p, off) for n, p, off in sizes]
        assert oracle.binary_concatenate(chunks).same(reference.binary_concatenate(chunks)), sizes


def test_nested_selection_restatement_against_reference_outputs():
    """The numpy restatement of the Selection kernels over non-primitive values (tests/nested.py) reproduces the committed
    outputs of the real reference (take / filter of (large_)binary, fixed_size_binary, decimal, (large_)list, map,
    fixed_size_list, struct, dictionary and null arrays; bounds errors), buffer for buffer."""
    import nested
    bad = nested.check_against_golden(nested.OracleNested())
    assert not bad, "\n".join(bad[:10])


def test_nested_selection_restatement_against_live_reference():
    import nested
    try:
        ref = nested.RefNested()
    except FileNotFoundError:
        pytest.skip("oracle/_ref not built (GPU box): the committed fixtures pin the restatement")
    rng = np.random.default_rng(123)
    orcl = nested.OracleNested()
    for kind in [nested.K_LARGE_UTF8, nested.K_FSB, nested.K_LIST, nested.K_LARGE_LIST, nested.K_MAP, nested.K_FSL, nested.K_STRUCT]:
        for trial in range(6):
            n = int(rng.integers(1, 80))
            v = nested.rand_node(rng, kind, n, float(rng.choice([0, 0.3])), 0 if kind == nested.K_FSL else int(rng.choice([0, 2])))
            for sel, flt, opt in ((nested.rand_indices(rng, n, 40, 0.2), 0, 1), (orc.random_arr(rng, orc.BOOL, n, 0.2), 1, int(trial % 2))):
                ok, msg = ref.select(v, sel, flt, opt).same(orcl.select(v, sel, flt, opt))
                assert ok, (nested.KIND_NAMES[kind], flt, msg)


def test_grouper_over_variable_width_and_dictionary_keys_against_reference_outputs():
    """The numpy restatement of GrouperImpl over binary / utf8 / large_* / dictionary key columns (alone and next to
    fixed-width ones, several batches) reproduces the committed ids and uniques of the real reference."""
    import nested
    bad = nested.check_groups_against_golden(nested.oracle_group)
    assert not bad, "\n".join(bad[:10])
