"""CPU: the C-ABI library loads and exports every symbol include/sosvo.h declares; the Python
signature table covers exactly that set.  No compute call is made here."""
import ctypes
import os

import pytest

from vo_single_camera_sos_amd import _lib


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported_and_bound():
    declared = _lib.declared_functions()
    assert declared, "header parse found nothing"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libsosvo.so lacks %s" % name
    assert sorted(_lib.SIGNATURES) == declared


def test_abi_version_and_loud_failure_without_gpu():
    lib = _lib.load()
    assert lib.sosvo_abi_version() >= 1
    import torch
    if not torch.cuda.is_available():
        from vo_single_camera_sos_amd.device import Context
        with pytest.raises(_lib.SosvoError):
            Context(0)


def test_generated_device_headers_are_current():
    """csrc/trig_core.h is the text of oracle/trig_core.h (tests/gen_device_headers.py): same arithmetic on both sides."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_device_headers", os.path.join(root, "tests", "gen_device_headers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(check=True), "run python tests/gen_device_headers.py"


def test_opencv_pattern_headers_and_c_entry_carry_the_data_file():
    """data/orb_bit_pattern_31.txt (OpenCV's bit_pattern_31_) == the generated oracle / device headers == what the C ABI
    hands out (sosvo_orb_bit_pattern_31: host memory, no context, no GPU)."""
    import ctypes
    import importlib.util
    import os
    import numpy as np
    from vo_single_camera_sos_amd import _lib, orb_pattern
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_orb_pattern", os.path.join(root, "scripts", "gen_orb_pattern.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(check=True), "run python scripts/gen_orb_pattern.py"
    buf = np.zeros((512, 2), dtype=np.int8)
    assert _lib.load().sosvo_orb_bit_pattern_31(ctypes.c_void_p(buf.ctypes.data)) == 0
    assert np.array_equal(buf, orb_pattern.opencv_pattern())
    assert _lib.load().sosvo_orb_bit_pattern_31(None) != 0
