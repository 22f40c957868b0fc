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


# ---- which extension kernel a GPU parity test runs on ----
# The automatic choice sends the small batches of the parity suite to the 64-lane one-read-per-wavefront kernel; the
# benchmark's batches run on the 8-lane groups (k_align_grp8, its _prim and _alt builds) behind the lane-per-read kernel.
# A test that takes the `kernels` fixture runs once per variant; the fixture sets the result-preserving kernel-selection
# options every Aligner of the test starts with and asserts afterwards (libmgx's launch counters) that the kernels it
# means to cover were the ones launched.
KERNEL_VARIANTS = {
    # name: (options, counters that must have moved, counters that must not have moved)
    "auto": ((), (), ()),
    # (map_pipe=2: k_map as the request / response machine also for these small batches — the automatic choice takes it from
    # 65536 chains on; "auto" keeps the one-step-per-lane machine, so both mapping kernels feed every parity test)
    "grp8": (("ext64=0", "lane=0", "map_pipe=2"), ("grp8_any",), ("ext64", "lane")),              # 8-lane groups, reads spread over wavefronts
    "grp8x8": (("ext64=0", "lane=0", "groups_per_wave=0", "map_pipe=2"), ("grp8_any",), ("ext64", "lane")),   # ... 8 reads per wavefront
    "lane": (("ext64=0", "lane=1", "groups_per_wave=0", "map_pipe=2", "seed_lane=1"), (), ("ext64",)),           # the lane-per-read kernels first (seeding and
                                                                                    # extension, where the configuration qualifies), the wave programs behind them
}


_LANE_SESSION = {"tests": 0, "launches": 0}


def pytest_sessionfinish(session, exitstatus):
    # a "lane" variant that never launched k_lane would be the other variants run twice
    if _LANE_SESSION["tests"] >= 20 and _LANE_SESSION["launches"] == 0:
        print("\nERROR: %d tests ran in the 'lane' kernel variant and the lane-per-read kernel was never launched" % _LANE_SESSION["tests"])
        session.exitstatus = 1


def _launch_counts():
    import ctypes as C
    from metagraph_amd import capi
    out = (C.c_uint64 * 5)()
    capi.lib().mgx_kernel_launch_counts(out)
    c = list(out)
    return {"grp8": c[0], "grp8_prim": c[1], "grp8_alt": c[2], "ext64": c[3], "lane": c[4], "grp8_any": c[0] + c[1] + c[2]}


@pytest.hookimpl(tryfirst=True, hookwrapper=True)
def pytest_runtest_makereport(item, call):
    outcome = yield
    rep = outcome.get_result()
    setattr(item, "rep_" + rep.when, rep)


@pytest.fixture(params=list(KERNEL_VARIANTS))
def kernels(request):
    from metagraph_amd import aligner
    opts, must, must_not = KERNEL_VARIANTS[request.param]
    before = _launch_counts()
    old = aligner.Aligner.default_options
    aligner.Aligner.default_options = tuple(opts)
    try:
        yield request.param
    finally:
        aligner.Aligner.default_options = old
    rep = getattr(request.node, "rep_call", None)
    if rep is None or not rep.passed:                        # (skipped or failed in its body: nothing to conclude from the counters)
        return
    after = _launch_counts()
    if request.param == "lane":
        # the lane-per-read kernel only takes the configurations lane_enabled() accepts (BASIC graphs, one alignment per query,
        # k <= 32 ...): not every test of the variant qualifies, but over the session many do (pytest_sessionfinish below)
        _LANE_SESSION["tests"] += 1
        _LANE_SESSION["launches"] += after["lane"] - before["lane"]
    for name in must:
        assert after[name] > before[name], "variant %s never launched %s" % (request.param, name)
    for name in must_not:
        assert after[name] == before[name], "variant %s launched %s" % (request.param, name)
