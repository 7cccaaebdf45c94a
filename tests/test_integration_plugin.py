"""INTEGRATION.md's reference-side binding, compiled (tests/integration/arrow_mi355x_plugin.cc): built in the build
container against the reference's own headers and linked against libamdarrow.so + the reference library of
oracle/ref_build.mk.  CPU: registration and dispatch under the reference's function names.  GPU: the reference's
CallFunction through that registry against the reference's own CPU kernels, bit for bit."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "integration", "_build", "plugin_test")


def _build():
    if os.path.isdir("/root/reference/cpp/src/arrow") and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libarrow_ref.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "integration")], stdout=subprocess.DEVNULL)
    if not os.path.exists(BIN):
        pytest.skip("plugin_test not built (needs the reference's headers: build container only)")


def test_plugin_registers_and_dispatches_under_the_reference_names():
    _build()
    r = subprocess.run([BIN, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout


@pytest.mark.gpu
def test_reference_callfunction_through_the_plugin_registry_matches_its_cpu_kernels():
    if not os.path.exists(BIN) or not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libarrow_ref.so")):
        pytest.skip("plugin_test / the reference library were not shipped to this box")
    r = subprocess.run([BIN, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout
