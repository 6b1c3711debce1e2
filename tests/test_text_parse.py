"""Input side, host checks: the decimal -> double converter of the HIP tokenizer (text_parse.hpp, compiled
for the host as well) against Python's float() — which is what the reference's readers return for every field
(src/mdapy/load_save.py:159-175) — and the generated table of powers of five against exact integer arithmetic."""
import ctypes
import random
import struct

import numpy as np

from mdapy_amd import _lib


def _conv(s):
    out = ctypes.c_double(0.0)
    b = s.encode()
    rc = _lib.lib().mdh_debug_parse_double(b, len(b), ctypes.byref(out))
    return rc, out.value


def _same(a, b):
    return struct.pack("<d", a) == struct.pack("<d", b)


def test_power_of_five_table_is_exact():
    L = _lib.lib()
    out = (ctypes.c_uint64 * 2)()
    for q in list(range(-342, 309, 7)) + [-342, -28, -27, -1, 0, 1, 27, 28, 55, 308]:
        if q >= 0:
            c = 5 ** q
            while c < (1 << 127):
                c *= 2
            while c >= (1 << 128):
                c //= 2
        else:
            p = 5 ** -q
            z = 0
            while (1 << z) < p:
                z += 1
            b = z + 127 if q >= -27 else 2 * z + 128
            c = (1 << b) // p + 1
            while c >= (1 << 128):
                c //= 2
        assert L.mdh_debug_text_pow5(q, out) == 0
        assert (int(out[0]) << 64) | int(out[1]) == c, q


def test_fields_round_exactly_like_float():
    rng = random.Random(5)
    cases = ["0", "-0", "0.0", "1", "-1.5", "3.615", "1e22", "1e23", "9007199254740993", "9007199254740992.5", "0.1", "1e-5",
             "123456789012345678", "1234567890123456789", "4.9e-324", "2.4703282292062327e-324", "2.4703282292062328e-324",
             "1.7976931348623157e308", "1.7976931348623159e308", "1e309", "1e-400", "2.2250738585072011e-308", "2.2250738585072014e-308",
             "+7.25", ".5", "5.", "1E5", "1e+5", "00012.500", "8.5e0", "0.000001", "123456.789e-3", "1.0000000000000002"]
    for _ in range(60000):
        kind = rng.randrange(6)
        if kind == 0:  # shortest round-trip spelling of a random double
            v = struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0]
            if v != v or v in (float("inf"), float("-inf")):
                continue
            cases.append(repr(v))
        elif kind == 1:
            cases.append("%.6g" % rng.uniform(-500, 500))
        elif kind == 2:
            cases.append("%.15f" % rng.uniform(-500, 500))
        elif kind == 3:
            cases.append("%de%d" % (rng.randrange(10 ** 19), rng.randrange(-340, 300)))
        elif kind == 4:  # exactly halfway between two doubles, and its neighbours
            m = rng.randrange(1 << 52, 1 << 53)
            cases.append(str(2 * m + 1) + rng.choice(["", "e-1", "e0"]))
        else:
            cases.append("%.17e" % rng.lognormvariate(0, 40))
    redo = 0
    for s in cases:
        rc, v = _conv(s)
        if rc == 1:
            redo += 1
            continue
        assert rc == 0, s
        assert _same(v, float(s)), (s, v, float(s))
    assert redo == 0  # at most 19 significant digits everywhere above: always decided


def test_long_fields_are_decided_or_handed_back():
    rng = random.Random(9)
    decided = 0
    for _ in range(20000):
        digits = "".join(rng.choice("0123456789") for _ in range(rng.randrange(20, 40)))
        s = digits[:3] + "." + digits[3:] + rng.choice(["", "e-7", "e12"])
        rc, v = _conv(s)
        assert rc in (0, 1)
        if rc == 0:
            decided += 1
            assert _same(v, float(s)), s
    assert decided > 19000
    assert _conv("1.00000000000000011102230246251565404")[0] == 1  # 36 digits, exactly between two doubles
    for s in ("9007199254740993.000000000000000001", "9007199254740993.0000000000000000000"):  # just above / exactly on a halfway point
        rc, v = _conv(s)
        assert rc == 1 or _same(v, float(s))
    for s, want in (("nan", 1), ("inf", 1), ("-inf", 1), ("abc", 2), ("1.5x", 1), ("1.5y", 2), ("", 2), ("1e", 2), ("0x10", 1), ("--1", 2)):
        assert _conv(s)[0] == want, s
