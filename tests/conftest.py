import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is test infrastructure: build it (g++ only) before any test runs
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def _have_gpu():
    if os.path.exists("/dev/kfd"):
        return True            # GPU hardware present: run the tests; a missing libmgx.so must fail loudly there
    try:
        from metagraph_amd import capi
        return capi.lib().mgx_device_count() > 0
    except OSError:
        return False


def pytest_collection_modifyitems(config, items):
    # a plain `pytest` on a box without libmgx.so or without a HIP device skips the GPU tests instead of erroring;
    # on a GPU box nothing is skipped, so a missing library still fails loudly there
    if any("gpu" in it.keywords for it in items) and not _have_gpu():
        skip = pytest.mark.skip(reason="needs libmgx.so and a HIP device")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)
