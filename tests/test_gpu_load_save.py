"""Input side on the GPU: the HIP tokenizer (csrc/text.hip through load_save.py) against the host readers —
same columns, same dtypes, same bits — on the reference's sample files, on a file built to be awkward, and on a generated
10 M-atom dump (timings go to gpurun_out/f2_reader.json)."""
import glob
import json
import os
import time

import numpy as np
import pytest

import mdapy_amd as mp
import mdapy_amd.load_save as LS

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _both(path, monkeypatch):
    with monkeypatch.context() as m:
        m.setattr(LS, "DEVICE_MIN_BYTES", 0)
        dev = LS.read_file(path)
    with monkeypatch.context() as m:
        m.setattr(LS, "have_gpu", lambda: False)
        host = LS.read_file(path)
    return dev, host


def _same(dev, host):
    assert dev[0].columns == host[0].columns and dev[2] == host[2]
    assert np.array_equal(dev[1].box, host[1].box) and np.array_equal(dev[1].origin, host[1].origin)
    for c in dev[0].columns:
        a, b = dev[0][c].to_numpy(), host[0][c].to_numpy()
        if a.dtype == object or b.dtype == object:
            assert list(a) == list(b), c
        else:
            assert a.dtype == b.dtype and a.tobytes() == b.tobytes(), c


def test_sample_files_device_equals_host(monkeypatch):
    files = sorted(glob.glob(os.path.join(HERE, "golden", "input_files", "*")))
    assert files
    for f in files:
        _same(*_both(f, monkeypatch))


def test_awkward_table(tmp_path, monkeypatch):
    rows = ["1 1 0.1 2.5e-3 -7 Cu", "2\t2   1e400  -0.0 +3.25\tZr\r", " 3 1 9007199254740993.0000000000000000001 4.9e-324 1.7976931348623157e308 Unobtainium",
            "4 2 nan inf -inf H extra fields here", "5 1 0.30000000000000004 123456789012345678901234567890 1e-330 He"]
    text = ("ITEM: TIMESTEP\n7\nITEM: NUMBER OF ATOMS\n5\nITEM: BOX BOUNDS pp pp pp\n0 10\n0 10\n0 10\nITEM: ATOMS id type x y z element\n"
            + "\n".join(rows))  # no newline after the last row
    p = tmp_path / "awkward.dump"
    p.write_bytes(text.encode())
    dev, host = _both(str(p), monkeypatch)
    assert list(dev[0]["element"].to_numpy()) == ["Cu", "Zr", "Unobtainium", "H", "He"]
    want = np.array([[float(t) for t in r.split()[2:5]] for r in rows])
    got = np.column_stack([dev[0][c].to_numpy() for c in "xyz"])
    assert got.tobytes() == want.tobytes()
    assert list(dev[0]["id"].to_numpy()) == [1, 2, 3, 4, 5] and dev[0]["id"].dtype == np.int32
    import gzip

    pz = tmp_path / "awkward.dump.gz"  # a compressed file has no known size: the pieces are concatenated in HBM
    with gzip.open(pz, "wb") as f:
        f.write(text.encode())
    devz, hostz = _both(str(pz), monkeypatch)
    _same(devz, hostz)
    assert np.column_stack([devz[0][c].to_numpy() for c in "xyz"]).tobytes() == want.tobytes()
    short = tmp_path / "short.dump"
    short.write_bytes(text.replace("5 1 0.30000000000000004 123456789012345678901234567890 1e-330 He", "5 1 0.3").encode())
    with monkeypatch.context() as m:
        m.setattr(LS, "DEVICE_MIN_BYTES", 0)
        with pytest.raises(ValueError, match="rows"):
            LS.read_file(str(short))


def test_ten_million_atom_dump(tmp_path):
    """generated 10 M-atom dump (17 significant digits per coordinate) -> System: columns resident in HBM, positions bit for
    bit what the host tokenizer (pandas, round-trip conversion == float()) and Python's float() give"""
    pd = pytest.importorskip("pandas")
    n = 10_000_000
    rng = np.random.default_rng(11)
    pos = rng.random((n, 3)) * 500.0 - 20.0
    p = str(tmp_path / "big.dump")
    t0 = time.perf_counter()
    with open(p, "w") as f:
        f.write(f"ITEM: TIMESTEP\n0\nITEM: NUMBER OF ATOMS\n{n}\nITEM: BOX BOUNDS pp pp pp\n-20 480\n-20 480\n-20 480\nITEM: ATOMS id type x y z\n")
        pd.DataFrame({"id": np.arange(1, n + 1, dtype=np.int32), "type": (np.arange(n) % 3 + 1).astype(np.int32), "x": pos[:, 0], "y": pos[:, 1],
                      "z": pos[:, 2]}).to_csv(f, sep=" ", header=False, index=False, float_format="%.17g")
    t_write = time.perf_counter() - t0
    size = os.path.getsize(p)
    import torch

    timings = []
    for rep in range(2):  # the second pass reads from the page cache and reuses the scratch buffers
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = mp.System(p)
        torch.cuda.synchronize(); timings.append(time.perf_counter() - t0)
    assert s.N == n and type(s.data["x"]._dev).__name__ == "HArray" and s.data["x"]._host_arr is None  # never left the GPU
    for k, c in enumerate("xyz"):
        assert s.data[c].to_numpy().tobytes() == pos[:, k].tobytes()  # %.17g round-trips a double exactly
    assert np.array_equal(s.data["id"].to_numpy(), np.arange(1, n + 1)) and np.array_equal(s.data["type"].to_numpy(), np.arange(n) % 3 + 1)
    # the same file through the host tokenizer, and a sample through float()
    t0 = time.perf_counter()
    host = pd.read_csv(p, sep=r"\s+", header=None, skiprows=9, names=["id", "type", "x", "y", "z"], float_precision="round_trip")
    t_host = time.perf_counter() - t0
    assert host["x"].to_numpy().tobytes() == s.data["x"].to_numpy().tobytes()
    with open(p) as f:
        lines = [next(f) for _ in range(9 + 50_000)][9:]
    assert np.array([[float(v) for v in ln.split()[2:]] for ln in lines]).tobytes() == pos[:50_000].tobytes()
    s.build_neighbor(2.0)  # the columns feed the kernels as they are
    out = {"atoms": n, "file_bytes": size, "device_reader_s": timings, "device_reader_GBps": [size / t / 1e9 for t in timings],
           "atoms_per_s": [n / t for t in timings], "pandas_round_trip_s": t_host, "pandas_GBps": size / t_host / 1e9, "write_s": t_write}
    os.makedirs(os.path.join(HERE, "..", "gpurun_out"), exist_ok=True)
    with open(os.path.join(HERE, "..", "gpurun_out", "f2_reader.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)


def test_many_host_converted_fields_and_truncated_tables(tmp_path, monkeypatch):
    """(1) Fields the device hands back — strings over 8 bytes, nan — by the thousand: patched in one batch, same frame as the
    host tokenizer; (2) more of them than the hand-back buffer holds (a long string on every one of 70 000 atoms): the reader
    falls back to the host tokenizer instead of failing; (3) a table cut off right after its header raises the reference's
    'expected N atom rows' error from both tokenizers, compressed or not."""
    import gzip
    import mdapy_amd._text as T
    head = "ITEM: TIMESTEP\n0\nITEM: NUMBER OF ATOMS\n{n}\nITEM: BOX BOUNDS pp pp pp\n0 50\n0 50\n0 50\nITEM: ATOMS id type x y z typelabel\n"
    rng = np.random.default_rng(3)

    def write(n, label_of, name):
        xyz = rng.random((n, 3)) * 50.0
        xyz = xyz.tolist()
        rows = [f"{i + 1} {1 + i % 2} {xyz[i][0]!r} {'nan' if i % 97 == 0 else repr(xyz[i][1])} {xyz[i][2]!r} {label_of(i)}" for i in range(n)]
        p = tmp_path / name
        p.write_text(head.format(n=n) + "\n".join(rows) + "\n")
        return str(p)

    few = write(5000, lambda i: "Copper-sixty-three" if i % 5 == 0 else "Cu", "few.dump")  # 1000 long strings + 52 nan
    dev, host = _both(few, monkeypatch)
    _same(dev, host)
    assert list(dev[0]["typelabel"].to_numpy()[:6]) == ["Copper-sixty-three", "Cu", "Cu", "Cu", "Cu", "Copper-sixty-three"]
    assert np.isnan(dev[0]["y"].to_numpy()[::97]).all()

    calls = []
    real = T.parse_table
    monkeypatch.setattr(T, "parse_table", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    many = write(70000, lambda i: "Zirconium-ninety", "many.dump")  # 70 000 > the 65 536-entry hand-back buffer
    dev, host = _both(many, monkeypatch)
    assert calls  # the device tokenizer was tried
    _same(dev, host)
    assert set(dev[0]["typelabel"].to_numpy()) == {"Zirconium-ninety"}
    monkeypatch.setattr(T, "parse_table", real)

    for gz in (False, True):
        p = tmp_path / ("cut.dump.gz" if gz else "cut.dump")
        data = head.format(n=1000).encode()
        p.write_bytes(gzip.compress(data) if gz else data)
        for gpu in (True, False):
            with monkeypatch.context() as m:
                m.setattr(LS, "DEVICE_MIN_BYTES", 0)
                if not gpu:
                    m.setattr(LS, "have_gpu", lambda: False)
                with pytest.raises(ValueError, match="expected 1000 atom rows"):
                    LS.read_file(str(p))
