"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/dsp_hip.h declares, and its
option defaults are sane.  No compute call is made without a GPU (dsp_create must refuse cleanly)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from dispatches_amd import hip_solver
    return hip_solver.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "dsp_hip.h")).read()
    declared = set(re.findall(r"\b(dsp_[a-z_]+)\s*\(", hdr))
    assert declared >= {"dsp_create", "dsp_solve", "dsp_spmv_step", "dsp_destroy", "dsp_strerror"}
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/dsp_hip.h but not exported"


def test_binding_mirrors_the_header(lib):
    """The ctypes structures are a field-for-field copy of the header's: same ABI version, same field names and order
    (a mismatch would make the library read the caller's structures with another layout)."""
    from dispatches_amd import hip_solver
    hdr = open(os.path.join(ROOT, "include", "dsp_hip.h")).read()
    assert int(re.search(r"#define DSP_VERSION (\d+)", hdr).group(1)) == hip_solver.ABI_VERSION == lib.dsp_version()
    for struct, cls in (("dsp_options", hip_solver.DspOptions), ("dsp_stats", hip_solver.DspStats),
                        ("dsp_lp_desc", hip_solver.DspLpDesc), ("dsp_batch", hip_solver.DspBatch),
                        ("dsp_wb_model", hip_solver.DspWbModel), ("dsp_wb_state", hip_solver.DspWbState),
                        ("dsp_loop_model", hip_solver.DspLoopModel), ("dsp_loop_state", hip_solver.DspLoopState)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        body = re.sub(r"\[[0-9\]\[]*\]", "", body)                     # array extents
        fields = [f for decl in body.split(";") for f in re.findall(r"\*?\s*([A-Za-z_][A-Za-z0-9_]*)\s*(?:,|$)", decl.strip().split(" ", 1)[-1] if decl.strip() else "")]
        fields = [f for f in fields if f not in ("const", "double", "int32_t", "int64_t")]
        assert fields == [f[0] for f in cls._fields_], (struct, fields, [f[0] for f in cls._fields_])


def test_default_options(lib):
    from dispatches_amd import hip_solver
    o = hip_solver.default_options()
    assert o.eps_rel == 1e-9 and o.eps_obj == 5e-7 and o.check_every == 0 and o.max_iter == 200000
    assert o.precision == 0 and o.polish_patience == 1024
    assert o.restart_artificial == 0.0 and o.pid_kp == 0.0          # automatic (per path, resolved in dsp_solve)
    assert o.kkt_every == 32 and o.kkt_gate == 16.0 and o.stall_rescue == 4000 and o.jump_rel == 3.0
    with pytest.raises(TypeError):
        hip_solver.default_options(not_an_option=1)
    assert lib.dsp_strerror(0) == b"ok" and b"invalid" in lib.dsp_strerror(-1)


def test_library_names_the_sources_it_was_built_from(lib, tmp_path):
    """dsp_source_hash() (ABI 9) is the SHA-256 prefix of csrc/*.hip, *.hpp and include/dsp_hip.h as they were at build time: the binding
    compares it with the tree next to it (a prebuilt .so that travelled with the tree must be the build of THESE sources - the driver's
    bench once timed a binary no hash tied to the code), build() rebuilds on a mismatch, the bench line prints it."""
    import shutil
    import __graft_entry__ as g
    from dispatches_amd import hip_solver
    have = lib.dsp_source_hash().decode()
    assert len(have) == 16 and have == hip_solver.source_hash() and not g._stale()
    assert hip_solver.default_options().eps_infeasible == 1e-6
    # a tree whose kernels differ by one byte has another hash; a tree without sources has none (nothing to compare with)
    root = tmp_path / "copy"
    shutil.copytree(os.path.join(ROOT, "dispatches_amd", "csrc"), root / "dispatches_amd" / "csrc", ignore=shutil.ignore_patterns(".*"))
    os.makedirs(root / "include")
    shutil.copy(os.path.join(ROOT, "include", "dsp_hip.h"), root / "include" / "dsp_hip.h")
    assert hip_solver.source_hash(str(root)) == have
    with open(root / "dispatches_amd" / "csrc" / "dsp_wave.hpp", "a") as fh:
        fh.write("\n")
    assert hip_solver.source_hash(str(root)) != have
    assert hip_solver.source_hash(str(tmp_path / "nowhere")) is None


def test_create_rejects_bad_input_without_gpu(lib):
    from dispatches_amd.hip_solver import DspLpDesc
    h = C.c_void_p()
    assert lib.dsp_create(None, 0, None, C.byref(h)) == -1
    rowptr = np.array([0, 2], np.int32); col = np.array([1, 0], np.int32); val = np.array([1.0, 2.0])
    d = DspLpDesc(2, 1, 2, rowptr.ctypes.data_as(C.POINTER(C.c_int32)), col.ctypes.data_as(C.POINTER(C.c_int32)),
                  val.ctypes.data_as(C.POINTER(C.c_double)))
    assert lib.dsp_create(C.byref(d), 0, None, C.byref(h)) == -1        # unsorted column indices


def test_solver_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import DspError, HipPdlpSolver
    solver = HipPdlpSolver()
    assert solver.available() is False
    bidder, model = scenarios.make_batch("nuclear_24h", 2, solver)
    with pytest.raises(DspError):
        solver.solve(model)


def test_integration_stub_matches_the_binding(lib):
    """INTEGRATION.md section 2 is what a maintainer copies: its structure definitions must be the current ABI's (round 2
    shipped a 6-field dsp_lp_desc stub next to a 7-field header: dsp_create would have read 8 bytes past the caller's struct)."""
    from dispatches_amd import hip_solver
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"## 2\..*?```python\n(.*?)```", doc, re.S).group(1)
    # the structure definitions of the stub, executed on their own (the rest of the block needs a GPU)
    classes = re.findall(r"^(class \w+\(C\.Structure\):.*?)(?=^\S)", block, re.S | re.M)
    assert classes, "section 2 no longer defines its structures"
    ns = {"C": C}
    for src in classes:
        exec(src, ns)                                                    # noqa: S102 - our own documentation
    checked = 0
    for name, cls in ns.items():
        ours = getattr(hip_solver, name, None)
        if isinstance(cls, type) and issubclass(cls, C.Structure) and ours is not None:
            assert [f[0] for f in cls._fields_] == [f[0] for f in ours._fields_], name
            assert C.sizeof(cls) == C.sizeof(ours), (name, C.sizeof(cls), C.sizeof(ours))
            checked += 1
    assert checked >= 1
    # every structure the stub does NOT define itself is imported from the binding, and the version it asserts is the header's
    for name in ("DspOptions", "DspBatch", "DspStats"):
        assert name in ns or re.search(r"from dispatches_amd\.hip_solver import[^\n]*\b%s\b" % name, block), name
    assert int(re.search(r"lib\.dsp_version\(\) == (\d+)", block).group(1)) == hip_solver.ABI_VERSION
    # every dsp_batch / dsp_lp_desc field the stub assigns exists
    for field in set(re.findall(r"\bb\.([a-z_0-9]+)\b", block)):
        assert field in {f[0] for f in hip_solver.DspBatch._fields_}, field
