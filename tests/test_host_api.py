"""CPU: host-side logic of the Python mirror that needs no device (name lookup, option
defaults, error codes as the reference raises them)."""
import pytest

import __graft_entry__ as ge

ge.load_package()
from arrow1_amd import _abi, compute  # noqa: E402


def test_unknown_function_is_key_error():
    with pytest.raises(_abi.AmdArrowError) as ei:  # registry.cc:62-64
        compute.call_function(None, "no_such_function", [])
    assert ei.value.code == 2 and "No function registered with name" in ei.value.msg


def test_hash_aggregates_cannot_be_called_directly():
    for name in ("hash_sum", "hash_count", "hash_mean"):
        with pytest.raises(_abi.AmdArrowError) as ei:  # function.cc:193-195
            compute.call_function(None, name, [None, None, None])
        assert ei.value.code == 10


def test_enum_values_match_the_reference():
    # FilterOptions::NullSelectionBehavior {DROP, EMIT_NULL}, SortOrder, CompareOperator order
    assert (_abi.DROP, _abi.EMIT_NULL) == (0, 1)
    assert (_abi.ASCENDING, _abi.DESCENDING) == (0, 1)
    assert [_abi.EQ, _abi.NE, _abi.GT, _abi.GE, _abi.LT, _abi.LE] == list(range(6))
    # arrow::Type::type ids, type_fwd.h:277-313
    assert (_abi.BOOL, _abi.UINT8, _abi.INT64, _abi.FLOAT, _abi.DOUBLE) == (1, 2, 9, 11, 12)


def test_context_requires_gpu():
    import torch
    from arrow1_amd import device
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        device.Context(0)
