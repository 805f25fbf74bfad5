"""On-disk formats: the product loaders (C ABI) and the
oracle loaders against the committed fixtures of tests/golden/.

The .modes / material expectations were produced by the REFERENCE's own
ModeData.h / ModalMaterial.h (oracle/_ref/ref_loaders, see make_golden.py);
the .fatcube bytes by the Python protobuf runtime from a descriptor built after
ffat_map.proto.  When oracle/_ref is present the reference binary is also run
live."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_loaders")


@pytest.fixture(scope="module")
def loaders():
    from openpbso_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    from openpbso_amd import loaders
    return loaders


def _dump():
    toks = open(os.path.join(G, "modes_small.dump.txt")).read().split()
    nd, nm = int(toks[0]), int(toks[1])
    vals = np.array([float.fromhex(t) for t in toks[2:]])
    return nd, nm, vals[:nm], vals[nm:].reshape(nm, nd)


def test_modes_product_loader_matches_reference_dump(loaders):
    nd, nm, om, md = _dump()
    got_om, got_md = loaders.read_modes(os.path.join(G, "modes_small_surf.modes"))
    assert got_md.shape == (nm, nd)
    assert np.array_equal(got_om, om) and np.array_equal(got_md, md)
    assert open(os.path.join(G, "modes_roundtrip.ok")).read().strip() == "identical"


def test_modes_oracle_loader_matches_reference_dump(oracle):
    nd, nm, om, md = _dump()
    l = oracle.lib()
    a, b = C.c_int(), C.c_int()
    po, pm = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    assert l.or_modes_read(os.path.join(G, "modes_small_surf.modes").encode(), C.byref(a), C.byref(b),
                           C.byref(po), C.byref(pm)) == 0
    assert (a.value, b.value) == (nd, nm)
    assert np.array_equal(np.ctypeslib.as_array(po, shape=(nm,)), om)
    assert np.array_equal(np.ctypeslib.as_array(pm, shape=(nm * nd,)).reshape(nm, nd), md)


def test_modes_missing_or_truncated_file(loaders, tmp_path):
    with pytest.raises(IOError):
        loaders.read_modes(str(tmp_path / "nope.modes"))
    p = tmp_path / "short.modes"
    p.write_bytes(open(os.path.join(G, "modes_small_surf.modes"), "rb").read()[:100])
    with pytest.raises(IOError):
        loaders.read_modes(str(p))


def test_num_modes_audible_matches_reference(loaders, oracle):
    _, _, om, _ = _dump()
    want = json.load(open(os.path.join(G, "audible.json")))
    for thr, n in want.items():
        assert loaders.num_modes_audible(om, 2500.0, float(thr)) == n
        assert oracle.lib().or_num_modes_audible(oracle._dp(om), om.size, 2500.0, float(thr)) == n
    assert loaders.num_modes_audible(np.zeros(0), 2500.0, 20000.0) == 0


def test_material_matches_reference(loaders, oracle):
    want = json.load(open(os.path.join(G, "materials.json")))
    for name, hexes in want.items():
        exp = [float.fromhex(h) for h in hexes]
        got = loaders.read_material(os.path.join(G, name))
        assert [got["density"], got["youngsModulus"], got["poissonRatio"], got["alpha"], got["beta"]] == exp
        out = np.zeros(5)
        assert oracle.lib().or_material_read(os.path.join(G, name).encode(), oracle._dp(out)) == 0
        assert out.tolist() == exp
    assert loaders.read_material(os.path.join(G, "does_not_exist.txt")) is None     # Read returns nullptr


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (reference tree absent)")
def test_live_reference_loaders_agree_with_fixtures():
    out = subprocess.run([REF, "modes_dump", os.path.join(G, "modes_small_surf.modes")], check=True,
                         capture_output=True, text=True).stdout
    assert out == open(os.path.join(G, "modes_small.dump.txt")).read()
    want = json.load(open(os.path.join(G, "materials.json")))
    for name, hexes in want.items():
        got = subprocess.run([REF, "material", os.path.join(G, name)], check=True, capture_output=True,
                             text=True).stdout.split()
        assert got == hexes


FIELDS = ["k", "cell_size", "center3", "center", "bbox_low", "bbox_top", "low_corners", "n_elements", "strides", "psi"]


def test_fatcube_product_and_oracle_decoders(loaders, oracle):
    meta = json.load(open(os.path.join(G, "fatcubes.json")))
    assert len(meta) == 3
    for name, m in meta.items():
        data = open(os.path.join(G, name), "rb").read()
        got = loaders.parse_fatcube(data)
        assert got["mode_id"] == m["mode_id"]          # modeid 0 is omitted on the wire (proto3 default)
        for f in FIELDS:
            assert np.array_equal(np.asarray(got[f], dtype=np.float64), np.asarray(m[f], dtype=np.float64)), (name, f)
        rc, om = oracle.fatcube_parse(data)
        assert rc == 0 and om.mode_id == m["mode_id"] and om.k == m["k"] and om.cell_size == m["cell_size"]
        assert np.array_equal(oracle.map_psi(om), np.asarray(m["psi"]))
        assert [list(r) for r in om.low_corners] == m["low_corners"]
        assert [list(r) for r in om.n_elements] == m["n_elements"]
        assert list(om.strides) == m["strides"] and list(om.bbox_low) == m["bbox_low"]
        assert list(om.bbox_top) == m["bbox_top"] and list(om.center3) == m["center3"]
        oracle.lib().or_ffat_free(C.byref(om))


def test_fatcube_survey_example_bytes(loaders):
    """SURVEY.md Appendix C example: k=3.5, modeid=7, psi=[[1,2,3]], cellsize=0.01."""
    import struct
    d = lambda x: struct.pack("<d", x)
    shells = b"\x09" + d(0.01)
    psi_vec = b"\x0a\x18" + d(1.0) + d(2.0) + d(3.0)
    psi = b"\x0a" + bytes([len(psi_vec)]) + psi_vec
    map3 = b"\x09" + d(3.5) + b"\x1a" + bytes([len(shells)]) + shells + b"\x2a" + bytes([len(psi)]) + psi + b"\x30\x07"
    top = b"\x0a" + bytes([len(map3)]) + map3
    assert top.hex().startswith("0a3409000000000000" + "0c40" + "1a0909" + "7b14ae47e17a843f" + "2a1c0a1a0a18")
    # the shell lacks lowcorners/n_elements/strides: the strict loader rejects it ...
    with pytest.raises(IOError):
        loaders.parse_fatcube(top)


def test_fatcube_malformed_inputs_are_rejected(loaders):
    good = open(os.path.join(G, "fat_uniform_mode5.fatcube"), "rb").read()
    for bad in (good[:-9], good[:40], b"\x0a\xff\xff\xff\xff\x0f" + good[:10], b"\xff" * 16):
        with pytest.raises(IOError):
            loaders.parse_fatcube(bad)
    # unknown fields are skipped (forward compatibility of proto3)
    extended = good + b"\x78\x01"       # field 15, varint 1 at top level
    assert loaders.parse_fatcube(extended)["mode_id"] == 5


def test_ffat_synthetic_geometry_equals_oracle_uniform_cube(oracle):
    """openpbso_amd.synth.uniform_cube_geometry (test-input generator) and the
    oracle's restatement of ResampleToUniformCube build the same cube."""
    from openpbso_amd import synth
    g = synth.uniform_cube_geometry((0.1, -0.2, 0.3), 0.01, 16)
    m = oracle.uniform_cube(3, 2.0, (0.1, -0.2, 0.3), 0.01, 16, np.zeros(6 * 256))
    assert np.array_equal(np.array([list(r) for r in m.low_corners]), g["low_corners"])
    assert list(m.bbox_low) == g["bbox_low"].tolist() and list(m.bbox_top) == g["bbox_top"].tolist()
    assert list(m.strides) == g["strides"].tolist()


def test_golden_audio_fixture_matches_oracle(oracle):
    z = np.load(os.path.join(G, "audio_c1.npz"))
    from openpbso_amd import synth
    s = oracle.Solver(z["lam"], synth.RHO, synth.ALPHA, synth.BETA)
    s.set_use_transfer(False)
    s.enqueue_force(z["data"])
    out = [s.step() for _ in range(4)]
    assert np.array_equal(np.concatenate([o[0] for o in out]), z["sound"])
    assert np.array_equal(np.array([o[1] for o in out]), z["qnorm"])


def test_obj_reader_and_area_weighted_vertex_normals(loaders, tmp_path):
    """<name>.tet.obj as tools/real_time_modal_sound.cpp:508-509 reads it: v / f records only, 1-based,
    negative and a/b/c face indices, polygons as triangle fans; per-vertex normals weighted by face area
    (libigl's default; the weighting itself is unpinned: libigl is not in the reference tree)."""
    # a unit cube: quads, mixed index forms, a comment, texture / normal records that must be ignored
    obj = """# cube
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0.5 0.5
vn 0 0 1
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
f 1 4 3 2
f 5/1 6/1 7/1 8/1
f 1//1 2//1 6//1 5//1
f -7 -6 -2 -3
f 3 4 8 7
g side
f 4/1/1 1/1/1 5/1/1 8/1/1
"""
    p = tmp_path / "cube.obj"
    p.write_text(obj)
    V, F, VN = loaders.read_obj(str(p))
    assert V.shape == (8, 3) and F.shape == (12, 3) and F.min() == 0 and F.max() == 7
    assert np.array_equal(F[0], [0, 3, 2]) and np.array_equal(F[1], [0, 2, 1])            # fan of the first quad
    assert np.array_equal(F[6], [1, 2, 6]) and np.array_equal(F[7], [1, 6, 5])            # -7 -6 -2 -3 with 8 vertices read
    fn = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]])
    want = np.zeros_like(V)
    for k in range(3):
        np.add.at(want, F[:, k], fn)
    want /= np.linalg.norm(want, axis=1, keepdims=True)
    assert np.allclose(VN, want, rtol=0, atol=1e-15)
    # outward faces of a cube: every vertex normal points away from the centre, and is a unit vector
    assert (np.einsum("ij,ij->i", VN, V - 0.5) > 0).all() and np.allclose(np.linalg.norm(VN, axis=1), 1.0)
    # area weighting: a sliver triangle barely turns the normal of a vertex it shares with a big one
    (tmp_path / "w.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1e-3\nf 1 2 3\nf 1 4 2\n")
    _, _, vn2 = loaders.read_obj(str(tmp_path / "w.obj"))
    assert vn2[0][2] > 0.999 and abs(vn2[0][1]) < 2e-3
    # a vertex that no face uses keeps a zero normal; malformed files are rejected
    (tmp_path / "lone.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 5 5 5\nf 1 2 3\n")
    assert np.array_equal(loaders.read_obj(str(tmp_path / "lone.obj"))[2][3], [0, 0, 0])
    for bad in ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n",
                "v 0 0 0\nv 1 0 0\nf 1 2\n", "v 0 0\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 -4\n"):
        (tmp_path / "bad.obj").write_text(bad)
        with pytest.raises(IOError):
            loaders.read_obj(str(tmp_path / "bad.obj"))
    with pytest.raises(IOError):
        loaders.read_obj(str(tmp_path / "missing.obj"))
