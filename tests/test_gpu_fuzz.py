"""Randomised message scripts: the engine's planner + kernels against the oracle's
ModalSolver::step for arbitrary interleavings of point / Gaussian / AR forces,
sustained start / end, clearAllForces, AR parameter updates, listener moves and
useTransfer toggles."""
import os

import numpy as np
import pytest

from openpbso_amd import capi, synth
from tests.scenarios import ObjSpec, force_ev, rel_errors, run_engine, run_oracle

pytestmark = pytest.mark.gpu


def random_script(rng, n_obj, n_modes, nb, with_maps):
    evs = []
    for oi in range(n_obj):
        sustained = False
        for b in range(nb):
            r = rng.random()
            n_msgs = 0 if r < 0.45 else (1 if r < 0.85 else 2)      # sometimes two messages in one frame (Q3)
            for _ in range(n_msgs):
                kind = rng.random()
                data = rng.standard_normal(n_modes[oi]) * 1e-3
                if sustained:
                    if kind < 0.15:
                        evs.append(force_ev(b, oi, force_type=2, end=True))
                        sustained = False
                    elif kind < 0.25:
                        evs.append(dict(t=b, obj=oi, kind="arprm", a=[float(rng.uniform(0.3, 0.8)), float(rng.uniform(0.0, 0.15))],
                                        sigma=float(rng.uniform(1e-3, 5e-3)), mu=float(rng.uniform(0.05, 0.3))))
                    else:
                        evs.append(force_ev(b, oi, data=data, force_type=2))
                else:
                    if kind < 0.45:
                        evs.append(force_ev(b, oi, data=data))
                    elif kind < 0.7:
                        w = float(rng.choice([0.0, 60.0, 400.0, 2500.0, 9000.0]))
                        evs.append(force_ev(b, oi, data=data, force_type=1, width=w))
                    elif kind < 0.8:
                        evs.append(force_ev(b, oi, data=data, force_type=2))            # plain AR force (lives forever)
                    elif kind < 0.9:
                        evs.append(force_ev(b, oi, data=data, force_type=2, start=True))
                        sustained = True
                    else:
                        evs.append(force_ev(b, oi, clear=True))
            if with_maps[oi]:
                if rng.random() < 0.5:
                    p = rng.standard_normal(3)
                    p = p / np.linalg.norm(p) * rng.uniform(0.2, 2.0) + 1e-3
                    evs.append(dict(t=b, obj=oi, kind="listener", pos=p))
                if rng.random() < 0.15:
                    evs.append(dict(t=b, obj=oi, kind="use_transfer", use=bool(rng.random() < 0.5)))
        if not with_maps[oi]:
            evs.append(dict(t=0, obj=oi, kind="use_transfer", use=False))
    return evs


def _legal(evs, objs, nb):
    """drop scripts the reference itself would abort on (clearAllForces while sustained leaves
    _sustainedForces set with an empty list: assert at modal_solver.h:223)"""
    try:
        run_oracle(objs, evs, nb)
        return True
    except AssertionError:
        return False


def _run_seed(seed, size_choices, engine_kw, projected_hits=False):
    rng = np.random.default_rng(1000 + seed)
    n_obj = int(rng.integers(1, 5))
    n_modes = [int(rng.choice(size_choices)) for _ in range(n_obj)]
    nb = int(rng.integers(6, 16))
    with_maps = [bool(rng.random() < 0.5) for _ in range(n_obj)]
    objs = []
    for oi in range(n_obj):
        lam = synth.eigenvalues(n_modes[oi], 5000 + 10 * seed + oi)
        objs.append(ObjSpec(lam, shapes=synth.mode_shapes(n_modes[oi], 6000 + seed + oi) if projected_hits else None,
                            maps=synth.ffat_maps(lam, 7000 + seed + oi, dim=4) if with_maps[oi] else None))
    evs = random_script(rng, n_obj, n_modes, nb, with_maps)
    if projected_hits:
        # replace the explicit modal data of about half of the messages by a vertex or face hit
        # (GetModalForceVertex / GetModalForceFace on the device)
        for e in evs:
            if e["kind"] == "force" and e["data"] is not None and rng.random() < 0.5:
                e["data"] = None
                e["vn"] = synth.unit_normals(1, int(rng.integers(1 << 30)))[0]
                if rng.random() < 0.5:
                    e["vid"] = int(rng.integers(0, synth.N_VERTS))
                else:
                    bary = rng.random(3)
                    e["vids"], e["coords"] = rng.integers(0, synth.N_VERTS, 3), bary / bary.sum()
    if not _legal(evs, objs, nb):
        pytest.skip("script trips a live assert of the reference")
    split = None
    if nb > 8:
        k = int(rng.integers(1, nb - 1))
        split = [k, nb - k]
    got = run_engine(objs, evs, nb, split=split, **engine_kw)
    want = run_oracle(objs, evs, nb)
    assert np.array_equal(got["emitted"], want["emitted"])
    mx, l2 = rel_errors(got["audio"], want["audio"])
    assert (mx <= 5e-4).all() and (l2 <= 1e-3).all(), (mx, l2)
    for a, w in zip(got["latest"], want["latest"]):
        assert np.array_equal(a, w)
    # qnorm: 5e-4 of the buffer's largest entry, plus 2e-6 of the object's peak over the run -- the ringing
    # left behind by a smooth (Gaussian) pulse is a 1e-4 residue of the response during the pulse, and
    # fp32 resolves it only relative to that response (seed 519)
    if engine_kw.get("qnorm") == capi.QNORM_OFF:
        return
    peak = {}
    for (oi, _), w in want["qnorm"].items():
        peak[oi] = max(peak.get(oi, 0.0), float(np.abs(w).max()))
    for key, w in want["qnorm"].items():
        tol = 5e-4 * max(np.abs(w).max(), 1e-30) + 2e-6 * peak[key[0]]
        assert np.abs(got["qnorm"][key] - w).max() <= tol, key


@pytest.mark.parametrize("seed", range(int(os.environ.get("PBSO_FUZZ_SEEDS", "200"))))
def test_random_scripts_match_oracle(seed, monkeypatch):
    if seed % 2:
        monkeypatch.setenv("PBSO_TEAM_WAVES", "16")     # whole objects as teams, as on a full chip
    # seeds 2, 3 of every four on the block kernel with the split-bf16 projection, the others on the default form
    # (PBSO_FORM; the block kernel with the f32 projection unless the run says otherwise)
    kw = dict(form=capi.FORM_BLOCK_BF16) if seed % 4 >= 2 else {}
    # K5 (launches cut along the time axis behind a scan of buffer-start states): forced on with 1, 2 or 7 buffers per chunk for
    # a third of the seeds -- whatever the script holds: dense profiles, clears, listener moves --, forced off for another third
    # (the walk in buffer order: K1b, or the pipeline kernel K1p for these small scenes), the engine's own choice for the rest
    if seed % 3 == 0:
        kw["time_chunks"] = [1, 2, 7][(seed // 3) % 3]
    elif seed % 3 == 1:
        kw["time_chunks"] = -1
    _run_seed(seed, [3, 40, 64, 100, 129, 300], kw)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PBSO_FUZZ_SHAPE_SEEDS", "60"))))
def test_random_scripts_random_engine_shapes(seed, monkeypatch):
    """the same scripts with on-device hit projection, objects stepped by several teams, every
    modes-per-lane setting, both qnorm modes and launches cut at random lengths"""
    rng = np.random.default_rng(77000 + seed)
    monkeypatch.setenv("PBSO_CHUNK_BUFFERS", str(int(rng.choice([1, 2, 5, 128]))))
    cap = int(rng.choice([0, 2, 16, 16]))            # 0: the engine's own choice (one wave per CU for engines this small)
    if cap:
        monkeypatch.setenv("PBSO_TEAM_WAVES", str(cap))
    # both oscillator-bank kernels: the block state-space form (K1b, the default) and the per-sample form (K1)
    form = int(rng.choice([capi.FORM_BLOCK, capi.FORM_BLOCK_BF16, capi.FORM_VELOCITY]))
    mpl = [0, 1, 2, 4, 8] if form != capi.FORM_VELOCITY else [0, 1, 2, 3, 4, 8]
    pick = int(rng.choice(mpl))
    if pick == 8 and form != capi.FORM_VELOCITY:
        pick = 4                                      # (eight modes per lane: the per-sample kernel only since round 6; the draw stays, so the scripts do)
    kw = dict(form=form, modes_per_lane=pick, qnorm=int(rng.choice([capi.QNORM_ALL, capi.QNORM_CLOSED])),
              time_chunks=int(rng.choice([0, -1, 1, 3])))
    _run_seed(seed + 100000, [5, 64, 200, 1100, 2100], kw, projected_hits=True)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PBSO_FUZZ_PIPE5_SEEDS", "48"))))
def test_random_scripts_five_role_pipeline_teams(seed):
    """round 5: the pipeline kernel's five-role teams (kernels_pipe.hip, iir_pipe5_kernel: one wave steps the state from increments
    that two others evaluated a buffer ahead, two project) pinned for EVERY launch of arbitrary scripts -- dense profiles,
    impulses, clears, listener moves with zero weights (the unscaled fallback), objects of one to five teams, an odd number of teams
    (a workgroup holds two), with and without qnorm rows, launches cut anywhere"""
    kw = dict(form=capi.FORM_BLOCK, modes_per_lane=1, bank_kernel=capi.BANK_PIPE, pipe_consumers=4,
              qnorm=capi.QNORM_ALL if seed % 2 else capi.QNORM_OFF)
    _run_seed(seed + 200000, [3, 64, 65, 130, 300], kw, projected_hits=bool(seed % 3 == 0))
