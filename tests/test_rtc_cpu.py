"""Run-time specialisation of the register-resident kernel (csrc/dsp_rtc.hpp), the part that needs no GPU: hiprtc compiles the
kernel sources for a shape no ahead-of-time table has, the code object is cached on disk, and a missing source tree is a
reported condition, not a crash (dsp_create then falls back to the padded / LDS-matrix ahead-of-time kernels)."""
import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture()
def lib(tmp_path, monkeypatch):
    from dispatches_amd import hip_solver
    monkeypatch.setenv("DSP_RTC_CACHE", str(tmp_path / "cache"))
    return hip_solver.load_library()


def _compile(lib, *shape):
    buf = C.create_string_buffer(4096)
    n = lib.dsp_rtc_compile_check(*shape, buf, 4096)
    return n, buf.value.decode()


def test_compile_an_untabulated_shape_and_hit_the_cache(lib, tmp_path):
    if not os.path.exists("/opt/rocm/lib/libhiprtc.so"):
        pytest.skip("no hiprtc in this environment")
    # wind+battery 20 h (cols / rows per lane 3 / 2) and the QP instantiation of a 30-h shape: neither is in DSP_MATREG_SHAPES
    for shape in ((3, 2, 0, 0x123, 0x34, 0), (4, 3, 0, 0x1234, 0x244, 1)):
        n, name = _compile(lib, *shape)
        assert n > 10000 and "pdlp_solve_kernel" in name, (n, name)
        assert f"Li{shape[0]}ELi{shape[1]}E" in name and name.endswith("EEvNS_9SolveArgsE")
    files = sorted(os.listdir(tmp_path / "cache"))
    assert len(files) == 4 and sum(f.endswith(".hsaco") for f in files) == 2, files
    before = {f: os.path.getmtime(tmp_path / "cache" / f) for f in files}
    n2, _ = _compile(lib, 3, 2, 0, 0x123, 0x34, 0)
    assert n2 > 10000 and {f: os.path.getmtime(tmp_path / "cache" / f) for f in files} == before      # served from the cache


def test_missing_sources_are_reported(lib, monkeypatch, tmp_path):
    monkeypatch.setenv("DSP_KERNEL_SRC", str(tmp_path / "nowhere"))
    n, why = _compile(lib, 3, 2, 0, 0x124, 0x34, 0)
    assert n == 0 and ("sources not found" in why or "hiprtc" in why), why
