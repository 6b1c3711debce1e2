"""Input side, host logic: header forms and column conventions of the reference's readers
(src/mdapy/load_save.py:66-198 dump, :653-863 XYZ, :610-650 .mp).  The atom table goes through the host tokenizer here
(no GPU); tests/test_gpu_load_save.py runs the same files through the HIP tokenizer and compares."""
import gzip

import numpy as np
import pytest

import mdapy_amd.load_save as LS


def _write(tmp_path, name, text):
    p = tmp_path / name
    if name.endswith(".gz"):
        with gzip.open(p, "wt") as f:
            f.write(text)
    else:
        p.write_text(text)
    return str(p)


DUMP_ORTHO = """ITEM: TIMESTEP
250
ITEM: NUMBER OF ATOMS
3
ITEM: BOX BOUNDS pp pp ff
-1.5 8.5
0 10
0 12.5
ITEM: ATOMS id type x y z vx element
1 1 0.5 1.25 2 -0.1 Cu
2 2 3.0000000000000004 4 5 0.25 Zr
3 1 6 7.125 8.0625 1e-3 Cu
"""


def test_dump_orthogonal_columns_and_box(tmp_path):
    for name in ("a.dump", "a.dump.gz"):
        fr, box, info = LS.read_file(_write(tmp_path, name, DUMP_ORTHO))
        assert info == {"timestep": 250}
        assert fr.columns == ["x", "y", "z", "id", "type", "vx", "element"]
        assert fr["id"].dtype == np.int32 and fr["type"].dtype == np.int32 and fr["x"].dtype == np.float64
        assert np.array_equal(fr["x"].to_numpy(), [0.5, 3.0000000000000004, 6.0]) and list(fr["element"].to_numpy()) == ["Cu", "Zr", "Cu"]
        assert np.array_equal(box.box, np.diag([10.0, 10.0, 12.5])) and np.array_equal(box.origin, [-1.5, 0, 0])
        assert list(box.boundary) == [1, 1, 0]


def test_dump_triclinic_forms_and_coordinate_variants(tmp_path):
    tilt = DUMP_ORTHO.replace("BOX BOUNDS pp pp ff", "BOX BOUNDS xy xz yz pp pp pp").replace("-1.5 8.5\n0 10\n0 12.5", "-1.5 9.5 1.0\n0 10 0.0\n0 12.5 0.5")
    _, box, _ = LS.read_file(_write(tmp_path, "t.dump", tilt))
    assert np.array_equal(box.box, [[10.0, 0, 0], [1.0, 9.5, 0], [0.0, 0.5, 12.5]]) and np.array_equal(box.origin, [-1.5, 0.0, 0.0])
    general = DUMP_ORTHO.replace("BOX BOUNDS pp pp ff", "BOX BOUNDS abc origin pp pp pp").replace("-1.5 8.5\n0 10\n0 12.5", "10 0 0 -1\n1 9 0 -2\n0 0.5 12 -3")
    _, box, _ = LS.read_file(_write(tmp_path, "g.dump", general))
    assert np.array_equal(box.box, [[10.0, 0, 0], [1, 9, 0], [0, 0.5, 12]]) and np.array_equal(box.origin, [-1, -2, -3])
    scaled = DUMP_ORTHO.replace("id type x y z vx element", "id type xs ys zs vx element").replace("0.5 1.25 2 ", "0.5 0.25 0 ")
    fr, box, _ = LS.read_file(_write(tmp_path, "s.dump", scaled))
    assert fr.columns[:3] == ["x", "y", "z"] and "xs" not in fr.columns
    assert np.array_equal([fr[c].to_numpy()[0] for c in "xyz"], box.origin + np.array([0.5, 0.25, 0.0]) @ box.box)
    unwrapped = DUMP_ORTHO.replace("id type x y z vx element", "id type xu yu zu vx element")
    fr, _, _ = LS.read_file(_write(tmp_path, "u.dump", unwrapped))
    assert fr.columns[:3] == ["x", "y", "z"] and np.array_equal(fr["y"].to_numpy(), [1.25, 4, 7.125])
    with pytest.raises(ValueError, match="no coordinate columns"):
        LS.read_file(_write(tmp_path, "n.dump", DUMP_ORTHO.replace("id type x y z vx element", "id type a b c vx element")))
    with pytest.raises(ValueError, match="multi-frame"):
        LS.read_file(_write(tmp_path, "m.dump", DUMP_ORTHO + DUMP_ORTHO))
    with pytest.raises(ValueError, match="atom rows"):
        LS.read_file(_write(tmp_path, "short.dump", DUMP_ORTHO.rsplit("\n", 2)[0] + "\n"))


def test_xyz_header_keys_aliases_and_classical(tmp_path):
    ext = ('2\nlattice="4 0 0 0 5 0 0 0 6" ORIGIN="1 2 3" Properties=species:S:1:pos:R:3:vel:R:3:q:R:1:tag:I:2 pbc="T F 1" energy=-3.5\n'
           "Cu 0 0.5 1 0.1 0.2 0.3 -1 7 8 trailing\nZr 2 2.5 3 0 0 0 1.5 9 10\n")
    fr, box, info = LS.read_file(_write(tmp_path, "e.xyz", ext))
    assert fr.columns == ["x", "y", "z", "element", "vx", "vy", "vz", "q", "tag_0", "tag_1"]
    assert fr["tag_1"].dtype == np.int32 and list(fr["tag_1"].to_numpy()) == [8, 10] and list(fr["element"].to_numpy()) == ["Cu", "Zr"]
    assert np.array_equal(box.box, np.diag([4.0, 5, 6])) and np.array_equal(box.origin, [1, 2, 3]) and list(box.boundary) == [1, 0, 1]
    assert info == {"energy": "-3.5"}
    classical = "3\njust a comment\nAr 0 0 1\nAr 2 0 1\nAr 0 4 1\n"
    fr, box, info = LS.read_file(_write(tmp_path, "c.xyz", classical))
    assert fr.columns == ["x", "y", "z", "element"] and list(box.boundary) == [0, 0, 0]
    assert np.allclose(np.diag(box.box), [2, 4, 1e-9]) and np.array_equal(box.origin, [0, 0, 1])
    twice = '1\nLattice="1 0 0 0 1 0 0 0 1" Properties=species:S:1:pos:R:3:force:R:3:forces:R:3\nH 0 0 0 1 2 3 4 5 6\n'
    fr, _, _ = LS.read_file(_write(tmp_path, "f.xyz", twice))
    assert fr.columns == ["x", "y", "z", "element", "fx", "fy", "fz", "forces_0", "forces_1", "forces_2"]
    with pytest.raises(ValueError, match="properties"):
        LS.read_file(_write(tmp_path, "bad.xyz", '1\nLattice="1 0 0 0 1 0 0 0 1"\nH 0 0 0\n'))


def test_mp_round_trip(tmp_path):
    pytest.importorskip("pyarrow")
    from mdapy_amd.box import Box
    from mdapy_amd.frame import Frame

    rng = np.random.default_rng(3)
    fr = Frame({"x": rng.random(50), "y": rng.random(50), "z": rng.random(50), "type": rng.integers(1, 3, 50).astype(np.int32),
                "element": np.array(["Cu", "Zr"], dtype=object)[rng.integers(0, 2, 50)]})
    box = Box(np.array([[3.0, 0, 0], [0.5, 4, 0], [0, 0, 5]]), [1, 0, 1], [0.25, -1, 2])
    p = str(tmp_path / "m.mp")
    LS.write_mp(p, fr, box, {"note": "hello"})
    got, gbox, info = LS.read_file(p)
    assert got.columns == fr.columns and info == {"note": "hello"}
    for c in fr.columns:
        assert list(got[c].to_numpy()) == list(fr[c].to_numpy())
    assert np.array_equal(gbox.box, box.box) and np.array_equal(gbox.origin, box.origin) and list(gbox.boundary) == [1, 0, 1]


def test_build_system_facade(tmp_path):
    """the reference's entry-point names (src/mdapy/load_save.py:356-410, 547-607) over this package's readers"""
    from mdapy_amd.load_save import BuildSystem

    frame, box = BuildSystem.from_array(np.arange(12.0).reshape(4, 3), [10.0, 11.0, 12.0])
    assert frame.columns == ["x", "y", "z"] and frame["y"].to_numpy().tolist() == [1.0, 4.0, 7.0, 10.0] and box.box[1, 1] == 11.0
    with pytest.raises(TypeError, match="numpy array"):
        BuildSystem.from_array([[0, 0, 0]], 5.0)
    with pytest.raises(ValueError, match="N x 3"):
        BuildSystem.from_array(np.zeros((3, 2)), 5.0)
    frame2, _ = BuildSystem.from_data({"x": np.array([1, 2]), "y": np.array([0, 0]), "z": np.array([3, 4]), "type": np.array([1, 2])}, 9.0)
    assert frame2["x"].to_numpy().dtype == np.float64 and "type" in frame2.columns
    with pytest.raises(ValueError, match="must contain z"):
        BuildSystem.from_data({"x": np.zeros(2), "y": np.zeros(2)}, 9.0)
    path = tmp_path / "two.xyz"
    path.write_text('2\nLattice="5 0 0 0 5 0 0 0 5" Properties=species:S:1:pos:R:3\nCu 0.5 0.5 0.5\nZr 1.5 2.5 3.5\n')
    got = BuildSystem.from_file(str(path))
    assert got[0].shape[0] == 2 and got[1].box[2, 2] == 5.0
    assert BuildSystem.from_file(str(path), format="XYZ")[0]["z"].to_numpy().tolist() == [0.5, 3.5]
    with pytest.raises(ValueError, match="not supported"):
        BuildSystem.from_file("frame.cfg")
    with pytest.raises(ValueError, match="outside its input side"):
        BuildSystem.from_file("POSCAR.poscar")
