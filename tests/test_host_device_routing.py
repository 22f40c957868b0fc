"""In-process multi-device host (metagraph_amd/host/hip_dbg_aligner.hpp, HipGraphSet): the worker -> device map, without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = """
#include "hip_dbg_aligner.hpp"
#include <cstdio>
int main() {
    using mgx::host::HipGraphSet;
    for (size_t D = 1; D <= 8; ++D) {
        size_t cnt[8] = {0};
        for (size_t w = 0; w < 64; ++w) { size_t d = HipGraphSet::device_of_worker(w, D); if (d >= D) return 1; ++cnt[d]; }
        for (size_t d = 0; d < D; ++d) if (cnt[d] < 64 / D) return 2;
    }
    std::puts("ok");
    return 0;
}
"""


def test_workers_are_spread_round_robin_over_the_devices(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    # (header-only use: nothing of libmgx.so is referenced by the routing function)
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "metagraph_amd", "host"), "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout
