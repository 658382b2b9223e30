"""The C-ABI library loads without a GPU, exports every symbol include/xpbd.h declares,
and fails loudly (never silently falls back) when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from constraint_solver_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "xpbd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xpbd_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert header_functions() == sorted(capi.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(capi.LIB_DIR, "libxpbd_hip.so"))
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.xpbd_abi_version() == 2


def test_rust_binding_text_declares_every_symbol():
    """constraint_solver_amd/ffi/xpbd_ffi.rs cannot be compiled here (no rustc); at least keep it complete."""
    text = open(os.path.join(ROOT, "constraint_solver_amd", "ffi", "xpbd_ffi.rs")).read()
    declared = set(re.findall(r"pub fn (xpbd_[a-z_0-9]+)\(", text))
    assert declared == set(header_functions())


def test_struct_layouts_match_header():
    assert C.sizeof(capi.Config) == 32
    assert capi.RIGID_DOUBLES * 8 == 304
    first = 0
    for name, (at, count) in capi.RIGID_FIELDS.items():   # contiguous, reference field order
        assert at == first, name
        first += count
    assert first == 38


def test_bad_config_is_rejected_before_touching_the_device():
    L = capi.hip_lib()
    h = C.c_void_p()
    cfg = capi.Config()
    L.xpbd_config_default(C.byref(cfg))
    assert (cfg.struct_size, cfg.device, cfg.mode, cfg.flags, cfg.block_size) == (32, 0, capi.MODE_FUSED, 0, 0)
    cfg.mode = 7
    assert L.xpbd_world_create(C.byref(h), C.byref(cfg)) == capi.E_INVALID
    assert b"mode" in L.xpbd_last_error()
    cfg.mode, cfg.block_size = 0, 96
    assert L.xpbd_world_create(C.byref(h), C.byref(cfg)) == capi.E_INVALID
    cfg.block_size, cfg.struct_size = 0, 8
    assert L.xpbd_world_create(C.byref(h), C.byref(cfg)) == capi.E_INVALID
    assert L.xpbd_world_create(None, None) == capi.E_INVALID
    assert h.value is None


def test_null_arguments_are_errors_not_crashes():
    L = capi.hip_lib()
    n = C.c_uint32()
    assert L.xpbd_world_step(None, 1 / 60, 20) == capi.E_INVALID
    assert L.xpbd_world_synchronize(None) == capi.E_INVALID
    assert L.xpbd_world_download_contacts(None, None, 0, C.byref(n)) == capi.E_INVALID
    assert L.xpbd_world_set_shapes(None, None, None, 0) == capi.E_INVALID
    assert L.xpbd_step_one(None, None, 0, 1 / 60, 1) == capi.E_INVALID
    assert L.xpbd_world_body_count(None) == 0
    assert L.xpbd_world_import_dynamic_rows(None, None, None, 0, None) == capi.E_INVALID
    assert L.xpbd_world_set_sat_schedule(None, 0) == capi.E_INVALID
    assert L.xpbd_world_history_push(None, None) == capi.E_INVALID
    assert L.xpbd_world_history_restore(None, 0) == capi.E_INVALID
    assert L.xpbd_world_history_truncate(None, 0) == capi.E_INVALID
    assert L.xpbd_world_history_length(None) == 0
    L.xpbd_world_destroy(None)   # no-op


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="this check is for the GPU-less build container")
def test_no_device_fails_loudly():
    with pytest.raises(capi.XpbdError) as e:
        capi.World()
    assert e.value.code == capi.E_NO_DEVICE
    with pytest.raises(capi.XpbdError):
        capi.step_one(np.zeros(38), np.zeros((8, 3)), 1 / 60, 4)
    with pytest.raises(capi.XpbdError):
        capi.selftest_div_sqrt(np.ones(4), np.ones(4))
