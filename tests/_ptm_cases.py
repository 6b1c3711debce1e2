"""Seeded PTM inputs shared by the CPU host test and the GPU parity test: (name, pos, box, boundary, structure, types, threshold)."""
import numpy as np

from mdapy_amd.build_lattice import lattice_positions


def _lat(kind, a, n):
    pos, box = lattice_positions(kind, a, n, n, n)
    box = np.asarray(box, float)
    return pos, (box if box.shape == (3, 3) else np.diag(box))


def _lonsdaleite(a, nx, ny, nz):
    """hexagonal diamond in its orthorhombic 8-atom cell (a, a*sqrt(3), c = a*sqrt(8/3)); returns pos, box, sublattice"""
    c, u = a * np.sqrt(8.0 / 3.0), 3.0 / 8.0
    hexb = np.array([[1 / 3, 2 / 3, 0.0], [2 / 3, 1 / 3, 0.5], [1 / 3, 2 / 3, u], [2 / 3, 1 / 3, 0.5 + u]])  # wurtzite sites
    kind = np.array([1, 1, 2, 2], np.int32)
    a1, a2 = np.array([a, 0, 0]), np.array([-a / 2, a * np.sqrt(3) / 2, 0])
    cart = hexb[:, :1] * a1 + hexb[:, 1:2] * a2 + hexb[:, 2:3] * np.array([0, 0, c])
    cell = np.concatenate([cart, cart + a1 + a2 * 0 + np.array([a / 2, a * np.sqrt(3) / 2, 0]) - a1])  # C-centred doubling
    lens = np.array([a, a * np.sqrt(3), c])
    cell = cell % lens
    kinds = np.concatenate([kind, kind])
    shifts = np.array([[i, j, k] for i in range(nx) for j in range(ny) for k in range(nz)], float) * lens
    pos = (cell[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
    return pos, np.diag(lens * np.array([nx, ny, nz])), np.tile(kinds, len(shifts))


def _graphene(d, nx, ny):
    """honeycomb sheet, bond length d, orthorhombic 4-atom cell, 20 A of vacuum along z (non-periodic)"""
    lx, ly = d * np.sqrt(3), 3 * d
    cell = np.array([[0, 0, 0], [0, d, 0], [lx / 2, 1.5 * d, 0], [lx / 2, 2.5 * d, 0]])
    kind = np.array([1, 2, 1, 2], np.int32)
    shifts = np.array([[i * lx, j * ly, 0] for i in range(nx) for j in range(ny)])
    pos = (cell[None] + shifts[:, None]).reshape(-1, 3) + np.array([0.1, 0.1, 10.0])
    return pos, np.diag([lx * nx, ly * ny, 20.0]), np.tile(kind, len(shifts))


def ptm_cases():
    rng = np.random.default_rng(20250928)
    pbc, out = (1, 1, 1), []
    everything = "fcc-hcp-bcc-ico-sc"
    for kind, a, n in (("fcc", 3.6, 6), ("bcc", 2.87, 7), ("hcp", 2.95, 6), ("sc", 2.5, 8)):
        pos, box = _lat(kind, a, n)
        for sig in (0.0, 0.02, 0.08, 0.2):
            p = pos + rng.normal(0, sig * a / 2 ** 0.5, pos.shape) if sig else pos.copy()
            out.append((f"{kind}_sig{sig}", p, box, pbc, everything, None, 0.1))
    pos, box = _lat("fcc", 3.6, 7)
    p = pos + rng.normal(0, 0.05, pos.shape)
    frac = np.round((pos / 3.6) % 1 * 2) / 2
    out.append(("fcc_random_binary", p, box, pbc, "default", rng.integers(1, 3, len(p)).astype(np.int32), 0.1))
    out.append(("fcc_L12", p, box, pbc, "default", np.where(np.all(frac == 0, axis=1), 2, 1).astype(np.int32), 0.1))
    out.append(("fcc_L12_au", p, box, pbc, "default", np.where(np.all(frac == 0, axis=1), 1, 2).astype(np.int32), 0.1))
    out.append(("fcc_L10", p, box, pbc, "default", np.where(frac[:, 2] == 0, 1, 2).astype(np.int32), 0.1))
    out.append(("fcc_ternary", p, box, pbc, "default", rng.integers(1, 4, len(p)).astype(np.int32), 0.1))
    pos, box = _lat("bcc", 2.87, 8)
    p = pos + rng.normal(0, 0.04, pos.shape)
    frac = np.round((pos / 2.87) % 1 * 2) / 2
    out.append(("bcc_B2", p, box, pbc, "default", np.where(np.all(frac == 0, axis=1), 1, 2).astype(np.int32), 0.1))
    out.append(("bcc_free_surfaces", p, box, (0, 0, 0), everything, None, 0.1))
    out.append(("bcc_slab", p, box, (1, 0, 1), "fcc-hcp-bcc", None, 0.1))
    out.append(("bcc_tight_threshold", p, box, pbc, "fcc-hcp-bcc", None, 0.02))
    out.append(("bcc_only_fcc", p, box, pbc, "fcc", None, 0.1))
    out.append(("bcc_no_threshold", p, box, pbc, "bcc,sc", None, 0.0))
    out.append(("random_gas", rng.random((4000, 3)) * 30.0, np.eye(3) * 30.0, pbc, everything, None, 0.5))
    out.append(("dilute_gas_few_neighbours", rng.random((60, 3)) * 30.0, np.eye(3) * 30.0, (0, 0, 0), everything, None, 0.5))
    pos, box = _lat("fcc", 3.6, 6)
    shear = np.array([[1, 0, 0], [0.15, 1, 0], [0.1, -0.07, 1]])
    out.append(("triclinic_fcc", (pos + rng.normal(0, 0.05, pos.shape)) @ shear, box @ shear, pbc, "default", None, 0.1))
    # two-shell types: diamond cubic / hexagonal, graphene (+ the zincblende / wurtzite / h-BN colourings)
    pos, box = _lat("diamond", 3.57, 5)
    sub = (np.round(pos / 3.57 * 4).astype(int).sum(1) % 4 != 0).astype(np.int32) + 1  # the two fcc sublattices
    for sig in (0.0, 0.03, 0.1):
        p = pos + rng.normal(0, sig, pos.shape) if sig else pos.copy()
        out.append((f"dcub_sig{sig}", p, box, pbc, "all", None, 0.1))
    out.append(("dcub_zincblende", pos + rng.normal(0, 0.03, pos.shape), box, pbc, "dcub-dhex", sub, 0.1))
    out.append(("dcub_only_simple_types", pos + rng.normal(0, 0.03, pos.shape), box, pbc, "default", None, 0.1))
    out.append(("dcub_free_surfaces", pos + rng.normal(0, 0.03, pos.shape), box, (0, 0, 0), "all", None, 0.1))
    hpos, hbox, hsub = _lonsdaleite(2.52, 6, 4, 4)
    for sig in (0.0, 0.03):
        p = hpos + rng.normal(0, sig, hpos.shape) if sig else hpos.copy()
        out.append((f"dhex_sig{sig}", p, hbox, pbc, "all", None, 0.1))
    out.append(("dhex_wurtzite", hpos + rng.normal(0, 0.03, hpos.shape), hbox, pbc, "dcub-dhex-graphene", hsub, 0.1))
    gpos, gbox, gsub = _graphene(1.42, 10, 6)
    for sig in (0.0, 0.03):
        p = gpos + rng.normal(0, sig, gpos.shape) if sig else gpos.copy()
        out.append((f"graphene_sig{sig}", p, gbox, (1, 1, 0), "all", None, 0.1))
    out.append(("graphene_hBN", gpos + rng.normal(0, 0.02, gpos.shape), gbox, (1, 1, 0), "graphene", gsub, 0.1))
    pos, box = _lat("fcc", 3.6, 6)
    out.append(("fcc_all_types", pos + rng.normal(0, 0.08, pos.shape), box, pbc, "all", None, 0.1))
    pos, box = _lat("bcc", 2.87, 7)
    out.append(("bcc_all_types", pos + rng.normal(0, 0.08, pos.shape), box, pbc, "all", None, 0.1))
    out.append(("random_gas_all_types", rng.random((3000, 3)) * 27.0, np.eye(3) * 27.0, pbc, "all", None, 0.5))
    pos, box = _lat("fcc", 3.6, 12)
    out.append(("fcc_melt_like", pos + rng.normal(0, 0.3, pos.shape), box, pbc, everything, None, 10.0))
    return out


def compare_ptm(out, ind, out_ref, ind_ref, tol=1e-6):
    """Bars: structure type, alloy ordering and matched-neighbour ids exact; rmsd / distance / quaternion 1e-6
    (the quaternion up to its overall sign)."""
    assert np.array_equal(out[:, 0], out_ref[:, 0]), "structure types differ"
    assert np.array_equal(out[:, 1], out_ref[:, 1]), "alloy orderings differ"
    assert np.array_equal(ind, ind_ref), "ptm_indices differ"
    assert np.allclose(out[:, 2:4], out_ref[:, 2:4], rtol=tol, atol=tol)
    dq = np.minimum(np.abs(out[:, 4:] - out_ref[:, 4:]).max(1), np.abs(out[:, 4:] + out_ref[:, 4:]).max(1))
    assert dq.max() <= tol
