"""CPU: the C-ABI library loads and exports every symbol include/sdrk.h declares;
the ctypes table in _ffi.py covers the same set.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from tests.conftest import REPO

HEADER = os.path.join(REPO, "include", "sdrk.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdrk_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    names = declared_symbols()
    for must in ("sdrk_plan_create", "sdrk_exec_host", "sdrk_exec_device", "sdrk_waterfall_create",
                 "sdrk_waterfall_append_rows", "sdrk_waterfall_read", "sdrk_synth_fill",
                 "sdrk_last_error", "sdrk_version", "sdrk_device_count", "sdrk_plan_destroy"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from sdr_iq_visualizer_amd import _ffi
    assert os.path.exists(_ffi.library_path()), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(_ffi.library_path())
    for name in declared_symbols():
        assert hasattr(handle, name), f"{name} declared in sdrk.h but not exported"


def test_ctypes_table_matches_header():
    from sdr_iq_visualizer_amd import _ffi
    table = sorted(name for name, _, _ in _ffi.SYMBOLS)
    assert table == declared_symbols()


def test_version_and_error_string_without_gpu():
    from sdr_iq_visualizer_amd import _ffi
    lib = _ffi.lib()
    header = int(re.search(r"#define\s+SDRK_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert lib.sdrk_version() == header == _ffi.ABI_VERSION == 500
    assert isinstance(lib.sdrk_last_error(), bytes)
    assert lib.sdrk_device_count() >= 0


def test_stale_library_is_refused_with_a_clear_message(monkeypatch):
    """A libsdrk.so built from an older header must not surface as an AttributeError on some newer symbol."""
    from sdr_iq_visualizer_amd import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "ABI_VERSION", 501)
    with pytest.raises(ImportError, match="reports ABI version 500, this package expects 501"):
        _ffi.lib()
    monkeypatch.setattr(_ffi, "ABI_VERSION", 500)
    assert _ffi.lib().sdrk_version() == 500


def test_product_path_fails_loudly_without_a_device():
    """No CPU fallback: with no GPU the numpy-facing API raises, it does not compute."""
    import numpy as np
    import sdr_iq_visualizer_amd as pkg
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.SdrkError):
        pkg.spectrum_db(np.zeros(4096, dtype=np.complex64))
    with pytest.raises(pkg.SdrkError):
        pkg.WaterfallBuffer(4096)
    with pytest.raises(pkg.SdrkError):
        pkg.process_frame(np.zeros(4096, dtype=np.complex64), 1e6, 2.4e9)
    with pytest.raises(pkg.SdrkError):
        pkg.pinned_empty((4, 4096), np.float32)
    with pytest.raises(pkg.SdrkError):
        with pkg.registered(np.zeros(4096, dtype=np.complex64)):
            pass
    from sdr_iq_visualizer_amd import features
    with pytest.raises(pkg.SdrkError):
        features.row_features(np.zeros(4096, dtype=np.float32))
    assert pkg.is_pinned(np.zeros(16, dtype=np.float32)) is False      # answers without a device: nothing is registered


def _build_c_smoke(tmp_path):
    import shutil
    import subprocess
    from sdr_iq_visualizer_amd import _ffi
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "c_abi_smoke")
    libdir = os.path.dirname(_ffi.library_path())
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(REPO, "include"),
                    os.path.join(REPO, "tests", "c_abi_smoke.c"), "-o", exe, "-L", libdir, "-lsdrk", "-lm",
                    "-Wl,-rpath," + libdir], check=True)
    return exe


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/sdrk.h compiles as strict C99 and a C program binds the library (no compute here)."""
    import subprocess
    exe = _build_c_smoke(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "sdrk 500" in out.stdout


@pytest.mark.gpu
def test_c_program_transforms_a_frame(tmp_path):
    import subprocess
    exe = _build_c_smoke(tmp_path)
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "at index 2148" in out.stdout
