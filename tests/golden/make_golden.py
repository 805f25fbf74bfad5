#!/usr/bin/env python3
"""Regenerates the committed fixtures in tests/golden/ (run from the repo root,
in the build container where /root/reference and the Python protobuf package
exist).  Fixtures are DATA: inputs and expected outputs.

  modes_small_surf.modes      byte fixture in ModeData's on-disk format
  modes_small.dump.txt        what the REFERENCE's ModeData<double>::read returns
                              for it (oracle/_ref/ref_loaders = reference headers
                              compiled where they lie), hex floats
  modes_roundtrip.ok          reference write(read(x)) == x byte for byte
  audible.json                reference numModesAudible for several thresholds
  material_*.txt / materials.json   ModalMaterial<double>::Read results (reference)
  fat_*.fatcube / fatcubes.json     .fatcube bytes written by the Python protobuf
                              runtime from a descriptor built after ffat_map.proto,
                              plus the field values that went in
  audio_c1.npz                oracle audio for the configs[0]-shaped case (fp64)
  kat_appendix_a.json         SURVEY.md Appendix A vectors (already in tests)
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref", "ref_loaders")


def ref(*args):
    return subprocess.run([REF, *args], check=True, capture_output=True, text=True).stdout


def make_modes():
    rng = np.random.default_rng(2024)
    n_dof, n_modes = 12, 7
    f = np.array([90.0, 450.0, 1800.0, 5200.0, 11000.0, 19999.0, 26000.0])
    omega2 = 2500.0 * (2 * np.pi * f) ** 2
    modes = rng.standard_normal((n_modes, n_dof))
    path = os.path.join(HERE, "modes_small_surf.modes")
    with open(path, "wb") as fh:
        fh.write(np.array([n_dof, n_modes], dtype=np.int32).tobytes())
        fh.write(omega2.astype(np.float64).tobytes())
        fh.write(modes.astype(np.float64).tobytes())
    open(os.path.join(HERE, "modes_small.dump.txt"), "w").write(ref("modes_dump", path))
    rt = os.path.join(HERE, "_rt.modes")
    ref("modes_roundtrip", path, rt)
    same = open(path, "rb").read() == open(rt, "rb").read()
    os.remove(rt)
    open(os.path.join(HERE, "modes_roundtrip.ok"), "w").write("identical\n" if same else "DIFFERENT\n")
    aud = {}
    for thr in [50.0, 90.0, 100.0, 2000.0, 19999.0, 20000.0, 22100.0, 30000.0]:
        aud[str(thr)] = int(ref("audible", path, "2500.0", repr(thr)))
    json.dump(aud, open(os.path.join(HERE, "audible.json"), "w"), indent=1)


def make_materials():
    cases = {
        "material_plain.txt": "2500 7.2e10 0.19 6.0 1e-7\n",
        "material_comments.txt": "# ceramic\n#density youngs poisson alpha beta\n2300.5 6.1E+10 0.25 12.5 3.0e-8\n# trailing\n",
        "material_short.txt": "# only three fields\n1000 2e9 0.3\n",
        "material_empty.txt": "# nothing else\n",
    }
    out = {}
    for name, text in cases.items():
        p = os.path.join(HERE, name)
        open(p, "w").write(text)
        out[name] = ref("material", p).split()
    json.dump(out, open(os.path.join(HERE, "materials.json"), "w"), indent=1)


from tests.fatcube_codec import proto_classes, encode_fatcube  # noqa: E402


def make_fatcubes():
    from openpbso_amd import synth
    cls = proto_classes()
    rng = np.random.default_rng(7)
    meta = {}
    # (a) two uniform cubes from the synthetic generator, modeid 0 (omitted on the wire) and 5
    lam = synth.eigenvalues(6, 99)
    maps = synth.ffat_maps(lam, 99, dim=4)
    for mid in (0, 5):
        m = maps[mid]
        name = f"fat_uniform_mode{mid}.fatcube"
        open(os.path.join(HERE, name), "wb").write(encode_fatcube(cls, m))
        meta[name] = {k: (np.asarray(v).tolist() if not np.isscalar(v) else v) for k, v in m.items()}
    # (b) a non-uniform map: different n_elements per face, off-centre, k from SURVEY's example
    ne = np.array([[3, 2], [3, 2], [2, 5], [2, 5], [5, 3], [5, 3]], dtype=np.int32)
    strides = np.concatenate([[0], np.cumsum(ne[:, 0] * ne[:, 1])[:-1]]).astype(np.int32)
    h = 0.02
    c = np.array([0.1, -0.2, 0.05])
    half = np.array([5, 3, 2]) * h / 2          # extents: x 5 cells, y 3 cells, z 2 cells
    low = np.zeros((6, 3))
    for dd in range(6):
        dk = dd // 2
        low[dd] = c - half
        low[dd, dk] = c[dk] + (half[dk] if dd % 2 == 0 else -half[dk])
    m = dict(mode_id=7, k=3.5, center3=c + 0.001, center=c, cell_size=h, low_corners=low, n_elements=ne,
             strides=strides, bbox_low=c - half, bbox_top=c + half,
             psi=np.abs(rng.standard_normal(int((ne[:, 0] * ne[:, 1]).sum()))) * 1e6)
    name = "fat_nonuniform_mode7.fatcube"
    open(os.path.join(HERE, name), "wb").write(encode_fatcube(cls, m))
    meta[name] = {k: (np.asarray(v).tolist() if not np.isscalar(v) else v) for k, v in m.items()}
    json.dump(meta, open(os.path.join(HERE, "fatcubes.json"), "w"))


def make_audio():
    from openpbso_amd import synth
    from oracle import oracle_py as orc
    lam = synth.eigenvalues(128, synth.seed_for(1, 0))
    shapes = synth.mode_shapes(128, synth.seed_for(1, 0))
    vn = synth.unit_normals(1, synth.seed_for(1, 0))[0]
    data = orc.modal_force_vertex(shapes, 0, vn)
    s = orc.Solver(lam, synth.RHO, synth.ALPHA, synth.BETA)
    s.set_use_transfer(False)
    s.enqueue_force(data)
    out = [s.step() for _ in range(4)]
    np.savez_compressed(os.path.join(HERE, "audio_c1.npz"), lam=lam, data=data,
                        sound=np.concatenate([o[0] for o in out]), qnorm=np.array([o[1] for o in out]))


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/ref_loaders missing: run `make -C oracle ref` where /root/reference exists")
    make_modes()
    make_materials()
    make_fatcubes()
    make_audio()
    print("fixtures written to", HERE)
