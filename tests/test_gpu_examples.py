"""The example scripts run end to end on the device and learn something (they double as integration tests of the
model classes on their intended workloads)."""
import importlib.util
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(name, argv):
    spec = importlib.util.spec_from_file_location("example_" + name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = sys.argv
    sys.argv = [name] + argv
    try:
        return mod.main()
    finally:
        sys.argv = old


def test_example_gmm():
    assert _run("gmm", ["--n", "20000", "--dim", "6", "--components", "4", "--iters", "20"]) > 0.9


def test_example_two_moons():
    ll0, ll1 = _run("two_moons", ["--n", "1500", "--experts", "6", "--iters", "25"])
    assert ll1 > ll0 + 0.5


def test_example_lorenz_lds():
    trace = _run("lorenz_lds", ["--series", "32", "--steps", "120", "--hidden", "6", "--iters", "6"])
    assert trace[-1] > trace[0]
