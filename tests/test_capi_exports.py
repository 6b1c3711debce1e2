"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol that include/mdapy_amd.h declares; compute entry points fail loudly without a device."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mdapy_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdh_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from mdapy_amd import _lib

    L = _lib.lib()
    names = _declared()
    assert len(names) >= 23
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/mdapy_amd.h but not exported"
    assert sorted(_lib.EXPORTS) == names, "ctypes signature table out of sync with the header"
    assert L.mdh_version() >= 100


def test_no_cpu_fallback_without_gpu():
    from mdapy_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    import mdapy_amd as mp

    s = mp.System(pos=np.random.default_rng(0).random((50, 3)) * 10, box=10.0)
    with pytest.raises(RuntimeError):
        s.build_neighbor(3.0)
    # the raw C ABI with host pointers fails loudly as well
    from mdapy_amd import _neighbor

    x = np.zeros(4); v = np.full((4, 2), -1, np.int32); d = np.zeros((4, 2)); nn = np.zeros(4, np.int32)
    with pytest.raises(RuntimeError, match="HIP error"):
        _neighbor.build_neighbor(x, x, x, np.eye(3) * 10, np.zeros(3), np.ones(3, np.int32), 3.0, v, d, nn, 1)


def test_argument_errors_map_to_python_exceptions():
    from mdapy_amd import _lib, _neighbor

    x = np.zeros(4); v = np.full((4, 2), -1, np.int32); d = np.zeros((4, 2)); nn = np.zeros(4, np.int32)
    with pytest.raises(ValueError):  # rc <= 0 is rejected before any device work
        _neighbor.build_neighbor(x, x, x, np.eye(3) * 10, np.zeros(3), np.ones(3, np.int32), -1.0, v, d, nn, 1)
    sing = np.array([[1.0, 1.0, 0.0], [2.0, 2.0, 0.0], [0.0, 0.0, 1.0]])  # singular triclinic box: src/box.h:185-186
    with pytest.raises(RuntimeError, match="volume of the box is zero"):
        _neighbor.build_neighbor(x, x, x, sing, np.zeros(3), np.ones(3, np.int32), 1.0, v, d, nn, 1)


def test_minimum_image_thresholds_are_exact():
    """DBox::tn — the kernels pick floor(d/L+0.5) by comparing d with these; each must be the exact step point."""
    from mdapy_amd import _lib

    L = _lib.lib()
    for length in (491.64, 36.15, 1.0, 7.3e-3, 123456.789):
        out = np.zeros(4)
        assert L.mdh_debug_image_thresholds(float(length), out.ctypes.data) == 0
        for k, t in enumerate(out):
            n = k - 1
            assert np.floor(t / length + 0.5) >= n
            assert np.floor(np.nextafter(t, -np.inf) / length + 0.5) < n
        assert out[0] < out[1] < out[2] < out[3]
        assert abs(out[1] + 0.5 * length) < 1e-9 * length and abs(out[2] - 0.5 * length) < 1e-9 * length
