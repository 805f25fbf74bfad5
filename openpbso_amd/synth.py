"""Deterministic synthetic inputs for the five BASELINE.json configurations
(SURVEY.md section 8(d)): no modal data ships with the reference, so tests,
bench.py and the harness generate objects, hits, listener paths and FFAT maps
here.  Pure input generation -- nothing in this file computes audio.
"""
import numpy as np

RHO, ALPHA, BETA = 2500.0, 6.0, 1e-7      # all modes under-damped for f <= 20 kHz
SPEED_OF_SOUND = 343.0
N_VERTS = 256
HIT_PROB = 0.233                          # Bernoulli per buffer ~ 20 hits/s at 86 buffers/s


def seed_for(config, obj):
    return 0x9B50 + 1000 * config + obj


def eigenvalues(n_modes, seed, f_lo=100.0, f_hi=18000.0):
    """lambda_m = rho (2 pi f_m)^2 with f_m log-uniform, ascending (ModeData order)."""
    rng = np.random.default_rng(seed)
    f = np.sort(np.exp(rng.uniform(np.log(f_lo), np.log(f_hi), n_modes)))
    return RHO * (2 * np.pi * f) ** 2


def mode_shapes(n_modes, seed, n_verts=N_VERTS):
    """ModeData::_modes, mode-major [n_modes][3 * n_verts]."""
    rng = np.random.default_rng(seed + 7919)
    return rng.standard_normal((n_modes, 3 * n_verts)) * 1e-3


def unit_normals(n, seed):
    rng = np.random.default_rng(seed + 104729)
    v = rng.standard_normal((n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def poisson_hits(n_buffers, seed, n_verts=N_VERTS, p=HIT_PROB):
    """per buffer: -1 (no hit) or the vertex id of a PointForce hit."""
    rng = np.random.default_rng(seed + 15485863)
    hit = rng.random(n_buffers) < p
    vid = rng.integers(0, n_verts, n_buffers)
    return np.where(hit, vid, -1)


def uniform_cube_geometry(center, cell_size, dim):
    """Cube-map geometry exactly as the reference's uniform-cube constructor
    (ResampleToUniformCube, ffat_solver.h:538-558)."""
    center = np.asarray(center, dtype=np.float64)
    low = np.zeros((6, 3))
    for dd in range(6):
        dk = dd // 2
        di, dj = (dk + 1) % 3, (dk + 2) % 3
        nml = +1 if dd % 2 == 0 else -1
        low[dd, dk] = center[dk] + nml * (dim // 2) * cell_size
        low[dd, di] = center[di] - (dim // 2) * cell_size
        low[dd, dj] = center[dj] - (dim // 2) * cell_size
    return {
        "cell_size": cell_size, "low_corners": low, "n_elements": np.full((6, 2), dim, dtype=np.int32),
        "strides": np.arange(6, dtype=np.int32) * dim * dim, "center": center.copy(),
        "center3": center.copy(), "bbox_low": low.min(axis=0), "bbox_top": low.max(axis=0),
    }


def ffat_maps(lam, seed, dim=16, cell_size=0.01, center=(0.0, 0.0, 0.0)):
    """One uniform cube map per mode: Psi ~ |N(0,1)| * 1e7 * k * 0.5 so that the
    transfer is ~1e7 at r = 0.5 m (the reference's unit transfer)."""
    rng = np.random.default_rng(seed + 32452843)
    omega = np.sqrt(np.asarray(lam) / RHO)
    maps = []
    for m, om in enumerate(omega):
        g = uniform_cube_geometry(center, cell_size, dim)
        g["mode_id"] = m
        g["k"] = om / SPEED_OF_SOUND
        g["psi"] = np.abs(rng.standard_normal(6 * dim * dim)) * 1e7 * g["k"] * 0.5
        maps.append(g)
    return maps


def listener_path(n_buffers, radius=0.5, steps_per_turn=86):
    """Circle of `radius` in a plane tilted off every axis (no zero direction
    component: SURVEY Q10), one position per buffer."""
    a = 2 * np.pi * np.arange(n_buffers) / steps_per_turn + 0.1234
    e1 = np.array([1.0, 0.35, 0.2])
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(e1, np.array([0.3, -0.2, 1.0]))
    e2 /= np.linalg.norm(e2)
    return radius * (np.cos(a)[:, None] * e1 + np.sin(a)[:, None] * e2)
