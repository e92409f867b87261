"""The C-ABI library loads and exports every symbol include/sige_hip.h declares.
No compute calls here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header(tuning=False):
    """include/sige_hip.h without comments; the `#ifdef SIGE_HIP_TUNING` block (measurement builds) only on request."""
    text = open(os.path.join(REPO, "include", "sige_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    block = re.compile(r"#ifdef SIGE_HIP_TUNING\n(.*?)#endif\n", re.S)
    assert len(block.findall(text)) == 1
    return block.sub(lambda m: m.group(1) if tuning else "", text)


def _declared(tuning=False):
    return sorted(set(re.findall(r"\b(sige_hip_[a-z0-9_]+)\s*\(", _header(tuning))))


@pytest.fixture(scope="module")
def lib():
    from sige_amd import build, hip

    build.build(verbose=False)
    assert os.path.isfile(hip.LIB_PATH)
    return ctypes.CDLL(hip.LIB_PATH)


def test_header_declares_something():
    names = _declared()
    assert len(names) >= 15 and "sige_hip_gather_f32" in names


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    from sige_amd import hip

    assert sorted(hip.EXPORTS) == _declared()


def _exported(path):
    import subprocess

    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


def test_product_library_has_no_dispatch_knobs(lib):
    """the round-4 `*_force_*` setters are gone; the product .so exports neither them nor the tuning pair."""
    from sige_amd import hip

    names = _exported(hip.LIB_PATH)
    bad = [n for n in names if n.startswith("sige_hip_") and ("_force_" in n or "tuning" in n or n.endswith("large_grid_nb1"))]
    assert not bad, bad
    assert not any("force" in n or "tuning" in n for n in _declared())
    assert not hip.lib().has_tuning
    with pytest.raises(RuntimeError, match="measurement build"):
        hip.tuning_set("conv_ksplit", 2)
    # every exported sige_hip_* symbol is declared in the header
    assert sorted(n for n in names if n.startswith("sige_hip_")) == _declared()


def test_tuning_build_exports_one_setter_and_getter():
    from sige_amd import build, hip

    path = build.build_tuning(verbose=False)
    names = {n for n in _exported(path) if n.startswith("sige_hip_")}
    assert sorted(names) == _declared(tuning=True)
    assert names - set(_declared()) == {"sige_hip_tuning_set", "sige_hip_tuning_get"}
    with hip.tuning_build() as L:
        assert L.has_tuning
        for key, default in (("conv_tile_mt", 0), ("conv_large_grid_nb1", -1), ("scatter_gather_form", 0)):
            assert hip.tuning_get(key) == default
        hip.tuning_set("conv_ksplit", 4)
        assert hip.tuning_get("conv_ksplit") == 4
        with pytest.raises(RuntimeError):
            hip.tuning_set("conv_ksplit", 99)       # out of the key's range
        assert L.sige_hip_tuning_set(999, 0) == -1    # unknown key
        assert L.sige_hip_tuning_get(999) == -2 ** 31
        hip.tuning_set("conv_tile_mt", 32)
    # the context restored the defaults and the product library
    assert not hip.lib().has_tuning
    with hip.tuning_build():
        assert hip.tuning_get("conv_ksplit") == 0 and hip.tuning_get("conv_tile_mt") == 0
    # the header's key numbers are the binding's
    keys = dict(re.findall(r"SIGE_HIP_TUNE_([A-Z0-9_]+) = (\d+)", _header()))
    assert {k.lower(): int(v) for k, v in keys.items() if k != "COUNT"} == hip.TUNE and int(keys["COUNT"]) == len(hip.TUNE)


def test_preload_without_a_gpu(lib):
    """sige_hip_preload needs a device: a clean status here, not a crash (the GPU tests check that it loads every unit)."""
    lib.sige_hip_preload.restype = ctypes.c_int
    assert lib.sige_hip_preload() in (-4, -3)


def test_version_and_error_strings(lib):
    lib.sige_hip_version.restype = ctypes.c_int
    lib.sige_hip_error_string.restype = ctypes.c_char_p
    assert lib.sige_hip_version() == 309
    assert lib.sige_hip_error_string(0) == b"ok"
    assert b"invalid" in lib.sige_hip_error_string(-1)


def test_argument_validation_without_a_gpu(lib):
    """Bad arguments are rejected before anything touches the device."""
    from sige_amd import hip

    h = hip.lib()
    assert h.sige_hip_gather_f32(None, 1, 1, 8, 8, 0, 6, None, 0, None, 0, 0, 0, 0, None, 0, 0, 0, 0, 0, 0, None, None) == -1
    assert h.sige_hip_gather_f32(None, 1, 1, 8, 8, 6, 6, None, 0, None, 0, 0, 0, 0, None, 0, 0, 0, 0, 7, 0, None, None) == -2
    assert h.sige_hip_reduce_mask_capacity(256, 256, 4, 4, 1, 1) == 65 * 65
    assert h.sige_hip_block_conv_packed_size(128, 128, 3, 3, 6, 6, 1, 1, 1) == 2 * 128 * 128 * 9  # both tile layouts
    assert h.sige_hip_block_conv_packed_size(128, 128, 5, 5, 8, 8, 1, 1, 1) == 0
    assert h.sige_hip_block_conv_packed_size(128, 128, 3, 3, 6, 6, 1, 1, 128) == 0


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "sige_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libsige_oracle" not in src, f


def _prototypes():
    """{name: (return class, [parameter classes])} parsed from include/sige_hip.h."""
    text = re.sub(r"//[^\n]*", "", _header())

    def cls(decl: str) -> str:
        decl = decl.strip()
        if "*" in decl:
            return "char*" if re.match(r"(const\s+)?char\b", decl) else "ptr"
        for key, name in (("size_t", "size"), ("int64_t", "i64"), ("double", "f64"), ("float", "f32"), ("int", "int")):
            if re.search(r"\b%s\b" % key, decl):
                return name
        raise AssertionError("unclassified parameter: %r" % decl)

    out = {}
    for ret, name, params in re.findall(r"([A-Za-z_][A-Za-z0-9_ ]*?[ *]+)(sige_hip_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        params = params.strip()
        plist = [] if params in ("", "void") else [cls(p) for p in params.split(",")]
        out[name] = (cls(ret), plist)
    return out


def test_python_binding_matches_the_header_parameter_by_parameter():
    """A ctypes signature that disagrees with the header only shows on the GPU (as garbage arguments): compare
    the class of every parameter and of the return value of every bound entry point with its prototype."""
    from sige_amd import hip

    names = {ctypes.c_void_p: "ptr", ctypes.c_int: "int", ctypes.c_size_t: "size", ctypes.c_int64: "i64",
             ctypes.c_float: "f32", ctypes.c_double: "f64", ctypes.c_char_p: "char*"}
    protos = _prototypes()
    assert sorted(protos) == _declared()
    for name, (res, args) in hip._SIGNATURES.items():
        want_res, want_args = protos[name]
        got_args = [names[a] for a in args]
        assert got_args == want_args, (name, [(i, g, w) for i, (g, w) in enumerate(zip(got_args, want_args)) if g != w],
                                       len(got_args), len(want_args))
        assert names[res] == want_res, name


def test_launch_plan_slot_table_without_a_gpu():
    """csrc/plan.hip: the host-only half of a launch plan -- slots, pointer bindings, begin / end state -- needs no device
    (no entry point is called while recording here, so nothing is launched)."""
    import ctypes

    from sige_amd import hip

    L = hip.lib()
    p = L.sige_hip_plan_create()
    assert p
    try:
        assert L.sige_hip_plan_new_slots(p, 3) == 0 and L.sige_hip_plan_new_slots(p, 2) == 3
        assert L.sige_hip_plan_set_slot(p, 4, 42) == 0 and L.sige_hip_plan_set_slot(p, 5, 1) != 0 and L.sige_hip_plan_set_slot(p, 0, -1) != 0
        assert L.sige_hip_plan_bind_ptr(p, 0x1000, 4) == 0 and L.sige_hip_plan_bind_ptr(p, 0x1000, 9) != 0 and L.sige_hip_plan_bind_ptr(p, None, 0) != 0
        arr = (ctypes.c_int32 * 5)()
        assert L.sige_hip_plan_get_slots(p, ctypes.cast(arr, ctypes.c_void_p), 5) == 5 and list(arr) == [0, 0, 0, 0, 42]
        assert L.sige_hip_plan_recording() == 0
        assert L.sige_hip_plan_begin(p, 0, 0) == 0 and L.sige_hip_plan_recording() == 1
        assert L.sige_hip_plan_begin(p, 1, 0) != 0          # (one recording per thread)
        assert L.sige_hip_plan_run(p, 0, None) != 0          # (not while recording)
        assert L.sige_hip_plan_end(p) == 0 and L.sige_hip_plan_recording() == 0
        assert L.sige_hip_plan_calls(p, 0) == 0 and L.sige_hip_plan_calls(p, 1) == 0 and L.sige_hip_plan_calls(p, 2) == -1
        assert L.sige_hip_plan_shape_bound(p) == 0 and L.sige_hip_plan_unbound(p) == 0
        assert L.sige_hip_plan_bind_const(p, 0x2000) == 0 and L.sige_hip_plan_bind_const(p, None) != 0
        assert L.sige_hip_plan_truncate(p, 0, 0) == 0 and L.sige_hip_plan_truncate(p, 0, 1) != 0 and L.sige_hip_plan_truncate(p, 2, 0) != 0
        assert L.sige_hip_plan_run(p, 1, None) == 0          # (an empty section)
    finally:
        assert L.sige_hip_plan_destroy(p) == 0


def test_plan_marks_count_arguments_on_unknown_pointers_without_a_gpu():
    """a count argument recorded next to a pointer the plan does not own used to replay the RECORDED count under
    every later mask, silently.  The hook runs before the entry point validates anything, so this needs no device: a
    gather call with a null tensor (-> EINVAL) still records; its index-list pointer is unknown -> unbound + shape bound; bound to a
    slot or registered as a constant -> neither; and a failed call can be truncated out of the plan."""
    from sige_amd import hip

    L = hip.lib()
    gather = L.sige_hip_gather_nhwc_f32.fn  # (the raw ctypes function: no device guard)

    def call(idx_ptr):
        return gather(None, 1, 4, 8, 8, 6, 6, idx_ptr, 5, None, 0, 0, None, 0, 0, 0, None, None)

    for bind, want_unbound in (("none", 1), ("slot", 0), ("const", 0)):
        p = L.sige_hip_plan_create()
        try:
            if bind == "slot":
                assert L.sige_hip_plan_new_slots(p, 1) == 0 and L.sige_hip_plan_bind_ptr(p, 0x3000, 0) == 0
            elif bind == "const":
                assert L.sige_hip_plan_bind_const(p, 0x3000) == 0
            assert L.sige_hip_plan_begin(p, 1, 0) == 0
            assert call(0x3000) == -1                      # (x is null: EINVAL -- after the hook stored the call)
            assert L.sige_hip_plan_calls(p, 1) == 1
            assert L.sige_hip_plan_unbound(p) == want_unbound and L.sige_hip_plan_shape_bound(p) == (1 if want_unbound else 0)
            assert L.sige_hip_plan_truncate(p, 1, 0) == 0 and L.sige_hip_plan_calls(p, 1) == 0   # what hip._Guarded does on an error status
            assert L.sige_hip_plan_end(p) == 0
            # the failed probe takes what it had marked with it -- the plan is not left shape bound by a call it no longer holds
            assert L.sige_hip_plan_unbound(p) == 0 and L.sige_hip_plan_shape_bound(p) == 0
        finally:
            assert L.sige_hip_plan_destroy(p) == 0
