"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the header
declares, the ctypes table matches the header, and argument errors are reported without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from dfnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "dfnet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dfn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dfnet_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.dfn_abi_version() == 1


def test_create_width_rules():
    """netwidth 128 = the register-resident kernels; any other even width is accepted (generic layer-by-layer path);
    odd widths and other depths / encodings are refused loudly."""
    lib = _lib.load()
    h = ctypes.c_void_p()
    d = _lib.NerfhDesc(8, 256, 10, 4, 10, 5, 2, 1000)
    assert lib.dfn_nerfh_create(ctypes.byref(d), ctypes.byref(h)) == 0
    assert lib.dfn_nerfh_destroy(h) == 0
    for bad in (_lib.NerfhDesc(8, 127, 10, 4, 10, 5, 2, 1000), _lib.NerfhDesc(6, 128, 10, 4, 10, 5, 2, 1000),
                _lib.NerfhDesc(8, 128, 8, 4, 10, 5, 2, 1000)):
        assert lib.dfn_nerfh_create(ctypes.byref(bad), ctypes.byref(h)) == -4 and b"dfn_nerfh_create" in lib.dfn_last_error()


def test_train_param_table_matches_state_dict_order():
    """dfn_nerfh_train_param_name() is the order of the params / grads pointer arrays: the state_dict order of both
    networks (models/nerfw.py:259-295) then the two embedding tables."""
    from dfnet_amd import synthetic as syn
    lib = _lib.load()
    n = lib.dfn_nerfh_train_param_count()
    names = [lib.dfn_nerfh_train_param_name(i).decode() for i in range(n)]
    cw, fw, _, _ = syn.nerfh_weights(0)
    assert names == ["coarse." + k for k in cw] + ["fine." + k for k in fw] + ["embedding_a.weight", "embedding_t.weight"]
    assert lib.dfn_nerfh_train_param_name(n) is None


def test_set_param_validation_and_commit_needs_all_params():
    lib = _lib.load()
    h = ctypes.c_void_p()
    d = _lib.NerfhDesc(8, 128, 10, 4, 10, 5, 2, 1000)
    assert lib.dfn_nerfh_create(ctypes.byref(d), ctypes.byref(h)) == 0
    a = np.zeros(128 * 63, np.float32)
    p = a.ctypes.data_as(ctypes.c_void_p)
    assert lib.dfn_nerfh_set_param(h, b"coarse.xyz_encoding_1.0.weight", p, a.size) == 0
    assert lib.dfn_nerfh_set_param(h, b"coarse.xyz_encoding_1.0.weight", p, 7) == -1
    assert b"expected 8064" in lib.dfn_last_error()
    assert lib.dfn_nerfh_set_param(h, b"coarse.nope", p, 1) == -1
    assert lib.dfn_nerfh_commit(h) == -3 and b"not set" in lib.dfn_last_error()
    # an uncommitted handle must refuse to run, loudly
    assert lib.dfn_mlp_coarse(h, 0, p, p, 1, 8, 0.0, 1.0, p, None) == -3
    assert lib.dfn_nerfh_destroy(h) == 0


def test_workspace_size_monotone():
    lib = _lib.load()
    a = lib.dfn_render_workspace_bytes(1000, 64, 128)
    b = lib.dfn_render_workspace_bytes(307200, 64, 128)
    assert 0 < a < b < (8 << 30)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdfnet_hip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.load()


def test_product_library_carries_no_bench_scaffolding():
    """The MFMA-rate probe of bench.py lives in its own helper library (tools/probe/libdfn_probe.so), and the shipped kernels carry
    no compile-time ablation switches: libdfnet_hip.so exports nothing named *probe*, the kernel sources have no DFN_ABL_ / DFN_DBG_ /
    DFN_CONV_ABL_ token."""
    import glob
    import subprocess
    so = os.path.join(ROOT, "dfnet_amd", "libdfnet_hip.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    assert "probe" not in syms.lower()
    for path in glob.glob(os.path.join(ROOT, "dfnet_amd", "csrc", "*.h*")):
        text = open(path).read()
        for tok in ("DFN_ABL_", "DFN_DBG_", "DFN_CONV_ABL_", "DFN_X3_NOSPREAD", "DFN_PRIO"):
            assert tok not in text, (os.path.basename(path), tok)
    probe = os.path.join(ROOT, "tools", "probe", "libdfn_probe.so")
    assert os.path.exists(probe)
    psyms = subprocess.run(["nm", "-D", "--defined-only", probe], capture_output=True, text=True, check=True).stdout
    assert "dfn_probe_mfma_rate" in psyms and "dfn_probe_last_error" in psyms


def test_round5_entries_refuse_bad_arguments_without_a_gpu():
    """The entry points added in round 5 (batched raygen / bicubic, the fused DFNet_dm loss block, the asynchronous range read, the
    pose orthogonalisation, the stand-alone conv weight gradient, the training modes) check their arguments before any device work:
    DFN_ERR_ARG with a message, on a box without a GPU too."""
    lib = _lib.load()
    P = ctypes.c_void_p
    one = ctypes.c_void_p(16)      # a non-null token: never dereferenced when another argument is refused first
    bad = [
        lib.dfn_raygen_frames(0, 4, 4, 1.0, one, one, one, None, None),
        lib.dfn_raygen_frames(2, 4, 4, -1.0, one, one, one, None, None),
        lib.dfn_raygen_frames_backward(2, 0, 4, 1.0, one, one, one, None),
        lib.dfn_upsample_bicubic_frames(None, 2, 4, 4, 3, 8, 8, 1, one, None),
        lib.dfn_upsample_bicubic_frames_backward(one, 0, 4, 4, 3, 8, 8, 0, one, None),
        lib.dfn_dm_loss_forward(one, one, 0, one, one, 12, None, 0.3, 0.2, 1.0, one, one, None),
        lib.dfn_dm_loss_forward(one, one, 8, one, one, 12, None, 0.3, 0.2, 1.0, None, one, None),
        lib.dfn_dm_loss_backward(one, one, 8, one, one, 0, 0.3, 0.2, 1.0, one, one, one, None, None),
        lib.dfn_pose_orthogonalize(None, 2, one, None),
        lib.dfn_nerfh_range_status_async(None, one, None),
        lib.dfn_conv_wgrad(None, one, 1, 8, 8, 64, 64, 3, one, None, one, 0, None),
    ]
    assert all(rc == -1 for rc in bad), bad
    assert lib.dfn_dm_loss_scratch_bytes() >= 2048
    assert b"dfn_" in lib.dfn_last_error()
    # training modes: 0 fused (fine operands as one f16 plane), 1 exact, 2 fused with hi | lo planes everywhere; anything else is refused
    h = ctypes.c_void_p()
    d = _lib.NerfhDesc(8, 128, 10, 4, 10, 5, 2, 1000)
    assert lib.dfn_nerfh_create(ctypes.byref(d), ctypes.byref(h)) == 0
    try:
        for mode in (0, 1, 2):
            assert lib.dfn_nerfh_set_train_mode(h, mode) == 0
        assert lib.dfn_nerfh_set_train_mode(h, 3) == -1
        assert lib.dfn_nerfh_range_status_async(h, None, None) == -1
    finally:
        lib.dfn_nerfh_destroy(h)
