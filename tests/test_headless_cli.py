"""File-based end-to-end path (SURVEY A8 + A10): a data directory in the
reference's layout (<n>.tet.obj, <n>_surf.modes, <n>_material.txt,
<n>_ffat_maps/*.fatcube + freq_threshold.txt) is read by the headless tool
(the reference's flags) on the GPU, and by the oracle's loaders on the CPU."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from openpbso_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "openpbso_amd", "pbso_headless")
B = 513


def make_data_dir(d, name="bowl", n_modes=40, n_verts=12, thr=9000.0):
    from tests.fatcube_codec import encode_fatcube, proto_classes
    os.makedirs(d / f"{name}_ffat_maps")
    rng = np.random.default_rng(5)
    lam = synth.eigenvalues(n_modes, 77, f_lo=150.0, f_hi=16000.0)
    shapes = rng.standard_normal((n_modes, 3 * n_verts)) * 1e-3
    with open(d / f"{name}.tet.obj", "w") as f:
        for v in rng.standard_normal((n_verts, 3)):
            f.write("v %.6f %.6f %.6f\n" % tuple(v))
        for i in range(n_verts):                       # a triangle strip that touches every vertex
            f.write("f %d %d %d\n" % (i + 1, (i + 1) % n_verts + 1, (i + 2) % n_verts + 1))
    with open(d / f"{name}_surf.modes", "wb") as f:
        f.write(np.array([3 * n_verts, n_modes], dtype=np.int32).tobytes())
        f.write(lam.tobytes())
        f.write(shapes.tobytes())
    (d / f"{name}_material.txt").write_text("# rho E nu alpha beta\n2500 7e10 0.2 6.0 1e-7\n")
    (d / f"{name}_ffat_maps" / "freq_threshold.txt").write_text(f"{thr}\n")
    cls = proto_classes()
    maps = synth.ffat_maps(lam, 78, dim=6)
    for m in maps:
        (d / f"{name}_ffat_maps" / f"mode_{m['mode_id']:03d}.fatcube").write_bytes(encode_fatcube(cls, m))
    return lam, shapes


def test_headless_tool_is_built():
    if not os.path.exists(EXE):
        import __graft_entry__
        __graft_entry__.build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_headless_directory_run_matches_oracle(tmp_path, oracle):
    d = tmp_path / "data"
    d.mkdir()
    lam, shapes = make_data_dir(d)
    nb = 8
    # the third hit carries no normal: the tool takes VN.row(vid) of the mesh (igl::per_vertex_normals, tools/...:509,607)
    from openpbso_amd import loaders
    V, F, VN = loaders.read_obj(str(d / "bowl.tet.obj"))
    fn = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]])           # double-area x unit normal of every face
    vn_np = np.zeros_like(V)
    for k in range(3):
        np.add.at(vn_np, F[:, k], fn)
    vn_np /= np.linalg.norm(vn_np, axis=1, keepdims=True)
    assert np.allclose(VN, vn_np, rtol=0, atol=1e-14) and V.shape == (12, 3) and F.shape == (12, 3)
    hits = [(0, 3, (0.2, -0.5, 1.0), "point"), (2, 7, (1.0, 0.0, 0.3), "gauss 400"), (5, 1, None, "point")]
    (tmp_path / "hits.txt").write_text("# buffer vid nx ny nz type\n" + "".join(
        (f"{b} {v} {n[0]} {n[1]} {n[2]} {t}\n" if n is not None else f"{b} {v} - {t}\n") for b, v, n, t in hits))
    path = synth.listener_path(nb)
    (tmp_path / "listener.txt").write_text("".join(f"{b} {float(p[0])!r} {float(p[1])!r} {float(p[2])!r}\n" for b, p in enumerate(path)))
    r = subprocess.run([EXE, "-d", str(d), "--hits", str(tmp_path / "hits.txt"), "--listener", str(tmp_path / "listener.txt"),
                        "--buffers", str(nb), "--out", str(tmp_path / "o.wav"), "--raw", str(tmp_path / "o.raw")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "object name: bowl" in r.stdout and "mesh: 12 vertices, 12 triangles" in r.stdout
    got = np.fromfile(tmp_path / "o.raw", dtype=np.float32).astype(np.float64)

    # ---- the same directory through the oracle's loaders
    l = oracle.lib()
    a, b = C.c_int(), C.c_int()
    po, pm = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    assert l.or_modes_read(str(d / "bowl_surf.modes").encode(), C.byref(a), C.byref(b), C.byref(po), C.byref(pm)) == 0
    om = np.ctypeslib.as_array(po, shape=(b.value,)).copy()
    md = np.ctypeslib.as_array(pm, shape=(b.value * a.value,)).reshape(b.value, a.value).copy()
    mat = np.zeros(5)
    assert l.or_material_read(str(d / "bowl_material.txt").encode(), oracle._dp(mat)) == 0
    thr = float(open(d / "bowl_ffat_maps" / "freq_threshold.txt").readline())
    n_aud = l.or_num_modes_audible(oracle._dp(om), om.size, mat[0], thr)
    assert 0 < n_aud < om.size and f"modes: {n_aud} of {om.size} audible" in r.stdout
    maps = []
    for fn in sorted(os.listdir(d / "bowl_ffat_maps")):
        if ".fatcube" in fn:
            m = oracle.OrFfatMap()
            assert l.or_fatcube_load(str(d / "bowl_ffat_maps" / fn).encode(), C.byref(m)) == 0
            maps.append(m)
    s = oracle.Solver(om, mat[0], mat[3], mat[4], n_modes=n_aud)
    s.read_ffat_maps(maps)
    want = []
    for bb in range(nb):
        s.compute_transfer(path[bb])
        for hb, v, n, t in hits:
            if hb == bb:
                vn = np.array(n) / np.linalg.norm(n) if n is not None else vn_np[v]
                f = oracle.make_force(oracle.GAUSSIAN, 400.0) if t.startswith("gauss") else oracle.make_force(oracle.POINT)
                s.enqueue_force(oracle.modal_force_vertex(md, v, vn, n_aud), f)
        want.append(s.step()[0])
    want = np.concatenate(want)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 5e-4 * np.abs(want).max()

    # the WAV holds what PaModalCallback would play: float32 mono, sound / 1e10
    raw = open(tmp_path / "o.wav", "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and struct.unpack("<HHI", raw[20:28]) == (3, 1, 44100)
    wav = np.frombuffer(raw[44:], dtype=np.float32)
    assert np.array_equal(wav, (got / 1e10).astype(np.float32))


@pytest.mark.gpu
def test_headless_dof_mismatch_is_reported(tmp_path):
    d = tmp_path / "data"
    d.mkdir()
    make_data_dir(d)
    with open(d / "bowl.tet.obj", "a") as f:
        f.write("v 0 0 0\n")
    r = subprocess.run([EXE, "-d", str(d), "--buffers", "1", "--out", str(tmp_path / "o.wav")], capture_output=True, text=True)
    assert r.returncode != 0 and "DOFs mismatch" in r.stderr


@pytest.mark.gpu
def test_headless_devices_flag_runs_the_scene_through_the_device_group(tmp_path):
    """--devices 0 --copies 3: three instances of the object on a group of one GPU (include/openpbso_amd.h "device group"),
    copy c one buffer later than copy c - 1, the WAV their on-device MIX -- equal to the sum of three single-engine runs of the
    tool with the scripts shifted by hand"""
    d = tmp_path / "data"
    d.mkdir()
    make_data_dir(d)
    nb = 7
    hits = [(0, 3, (0.2, -0.5, 1.0)), (2, 7, (1.0, 0.0, 0.3)), (3, 1, (0.0, 1.0, 0.0))]
    path = synth.listener_path(nb)

    def scripts(shift, tag):
        (tmp_path / f"h{tag}.txt").write_text("".join(f"{b + shift} {v} {n[0]} {n[1]} {n[2]} point\n" for b, v, n in hits))
        (tmp_path / f"l{tag}.txt").write_text("".join(f"{b + shift} {float(p[0])!r} {float(p[1])!r} {float(p[2])!r}\n" for b, p in enumerate(path)))
        return ["--hits", str(tmp_path / f"h{tag}.txt"), "--listener", str(tmp_path / f"l{tag}.txt")]

    r = subprocess.run([EXE, "-d", str(d)] + scripts(0, "g") + ["--buffers", str(nb), "--devices", "0", "--copies", "3",
                        "--out", str(tmp_path / "g.wav"), "--raw", str(tmp_path / "g.raw")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "3 copies x" in r.stdout and "ranks own [0, 3)" in r.stdout
    got = np.fromfile(tmp_path / "g.raw", dtype=np.float32).astype(np.float64)
    want = np.zeros(nb * B)
    for c in range(3):
        r1 = subprocess.run([EXE, "-d", str(d)] + scripts(c, f"s{c}") + ["--buffers", str(nb), "--out", str(tmp_path / "s.wav"),
                             "--raw", str(tmp_path / f"s{c}.raw")], capture_output=True, text=True)
        assert r1.returncode == 0, r1.stderr
        want += np.fromfile(tmp_path / f"s{c}.raw", dtype=np.float32).astype(np.float64)
    assert np.abs(want).max() > 0 and np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
