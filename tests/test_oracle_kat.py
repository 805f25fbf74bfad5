"""Pins the fp64 CPU oracle (oracle/pbso_oracle.c).

Known-answer vectors: SURVEY.md Appendix A -- produced at survey time from the
reference's UNMODIFIED modal_integrator.h / forces.h (g++ 11.4, x86-64).  The
reference has no tests or golden files of its own, so
these plus the independent formulations below are the whole pin.
"""
import ctypes as C
import os
import subprocess

import numpy as np

RHO, ALPHA, BETA = 2500.0, 6.0, 1e-7
H = 1.0 / 44100
F_KAT = np.array([440.0, 1234.5, 5000.0, 12000.0])
LAM_KAT = RHO * (2 * np.pi * F_KAT) ** 2

# Step(Q=1) once then Step() four times
KAT_Q = np.array([
    [0.5266010819937037, 0.51039217436299333, 0.41291242249056054, 0.094622460849102183],
    [1.0510527133780474, 1.0048985900637348, 0.62425062569292367, -0.026031856899775965],
    [1.5712949838451502, 1.4682669804446351, 0.53182334144032972, -0.086236112395414205],
    [2.0852848349498698, 1.8862099098215115, 0.18125254980729233, 0.049419604540691942],
    [2.5910040834143686, 2.2458453089756714, -0.25654042851414532, 0.071524033702594955],
])


def test_integrator_trajectory_kat(oracle):
    l = oracle.lib()
    lam = LAM_KAT.copy()
    it = l.or_integrator_build(RHO, oracle._dp(lam), 4, ALPHA, BETA, H, -1)
    ones = np.ones(4)
    rows = [np.ctypeslib.as_array(l.or_integrator_step(it, oracle._dp(ones)), shape=(4,)).copy()]
    for _ in range(4):
        rows.append(np.ctypeslib.as_array(l.or_integrator_step_free(it), shape=(4,)).copy())
    l.or_integrator_free(it)
    got = np.array(rows)
    # vectors were transcribed with 17 significant digits: exact to 1 ulp
    np.testing.assert_allclose(got, KAT_Q, rtol=4e-16, atol=0)


def test_coeffs_kat_and_identities(oracle):
    c1, c2, c3 = oracle.iir_coeffs(LAM_KAT, RHO, ALPHA, BETA)
    np.testing.assert_allclose(c3, KAT_Q[0], rtol=4e-16)
    np.testing.assert_allclose(c1 * c3, KAT_Q[1], rtol=4e-16)     # q_1 = c1*c3
    # closed form: eps = sqrt(-c2), theta = acos(c1/(2 eps))
    om = 2 * np.pi * F_KAT
    xi = 0.5 * (ALPHA / om + BETA * om)
    a = 2 * xi * om
    np.testing.assert_allclose(np.sqrt(-c2), np.exp(-a * H / 2), rtol=1e-14)
    theta = H * np.sqrt(om ** 2 - a ** 2 / 4)
    np.testing.assert_allclose(c1 / (2 * np.sqrt(-c2)), np.cos(theta), rtol=1e-13)


def test_gaussian_force_kat(oracle):
    f = oracle.make_force(oracle.GAUSSIAN, 100.0)
    assert f.width_samples == 4 and f.center == 18
    buf = np.zeros(513)
    assert oracle.force_add(f, buf)
    np.testing.assert_allclose(buf[:5], [4.0065297392951069e-05, 0.00011961288358102437,
                                         0.00033546262790251185, 0.00088382630693504996,
                                         0.0021874911181828851], rtol=4e-16)
    # count = 513 >= 10*w = 40 -> dead on the next Add
    assert not oracle.force_add(f, buf)


def test_gaussian_zero_width_rejected(oracle):
    f = oracle.make_force(oracle.GAUSSIAN, 0.0)
    buf = np.zeros(513)
    assert not oracle.force_add(f, buf)
    assert not buf.any()


def test_gaussian_long_force_spans_buffers(oracle):
    # width 2000 us -> w = 88 samples, lives while count < 880 -> two buffers
    f = oracle.make_force(oracle.GAUSSIAN, 2000.0)
    assert f.width_samples == 88 and f.center == 396
    b0, b1, b2 = np.zeros(513), np.zeros(513), np.zeros(513)
    assert oracle.force_add(f, b0) and oracle.force_add(f, b1)
    assert not oracle.force_add(f, b2)
    i = np.arange(1026)
    ref = np.exp(-0.5 * ((i - 396) / 88.0) ** 2)
    np.testing.assert_allclose(np.concatenate([b0, b1]), ref, rtol=1e-14)


def test_point_force_kat(oracle):
    f = oracle.make_force(oracle.POINT)
    buf = np.zeros(513)
    assert oracle.force_add(f, buf)
    assert buf[0] == 1.0 and not buf[1:].any()
    assert not oracle.force_add(f, buf)


def test_ar_force_kat(oracle):
    f = oracle.make_force(oracle.AR)
    buf = np.zeros(513)
    assert oracle.force_add(f, buf)
    np.testing.assert_allclose(buf[:3], [0.14181949063947041, 0.14025017046519853,
                                         0.14162169350524792], rtol=4e-16)


def test_rng_matches_libstdcxx(oracle):
    """or_rng_normal() vs the real std::default_random_engine +
    std::normal_distribution<double> of this image's libstdc++ (the dependency
    the reference's forces.h:71-72 uses)."""
    exe = os.path.join(os.path.dirname(oracle.__file__), "stdrandom_check")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe), "stdrandom_check"], check=True)
    n = 5000
    out = subprocess.run([exe, str(n)], check=True, capture_output=True, text=True).stdout.split()
    want = np.array([float.fromhex(x) for x in out])
    r = oracle.OrRng()
    oracle.lib().or_rng_init(C.byref(r))
    got = np.array([oracle.lib().or_rng_normal(C.byref(r)) for _ in range(n)])
    # bit-exact: same engine draws, same rejection decisions, same libm
    assert np.array_equal(got, want)


def test_ar_set_param_keeps_rng_and_index(oracle):
    f = oracle.make_force(oracle.AR)
    buf = np.zeros(513)
    oracle.force_add(f, buf)
    x_before, idx_before = f.rng.x, f.buf_idx
    a = np.array([0.5, 0.2])
    oracle.lib().or_force_ar_set_param(C.byref(f), oracle._dp(a), 0.01, 0.3)
    assert f.rng.x == x_before and f.buf_idx == idx_before
    assert list(f.buf) == [0.0, 0.0, 0.0] and f.mu == 0.3


def test_iir_equals_lfilter(oracle):
    """Independent formulation: the bank is M parallel 2-pole filters."""
    from scipy.signal import lfilter
    rng = np.random.default_rng(1)
    f = np.sort(np.exp(rng.uniform(np.log(100), np.log(18000), 16)))
    lam = RHO * (2 * np.pi * f) ** 2
    c1, c2, c3 = oracle.iir_coeffs(lam, RHO, ALPHA, BETA)
    sol = oracle.Solver(lam, RHO, ALPHA, BETA)
    sol.set_use_transfer(False)
    S = rng.standard_normal(16)
    sol.enqueue_force(S)
    sound = np.concatenate([sol.step()[0] for _ in range(3)])
    drive = np.zeros(3 * 513)
    drive[0] = 1.0
    ref = np.zeros_like(drive)
    for m in range(16):
        ref += 1e7 * lfilter([c3[m] * S[m]], [1.0, -c1[m], -c2[m]], drive)
    np.testing.assert_allclose(sound, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())


def test_impulse_response_closed_form(oracle):
    """q_k = c3 * eps^k * sin((k+1) theta) / sin(theta)."""
    lam = LAM_KAT.copy()
    c1, c2, c3 = oracle.iir_coeffs(lam, RHO, ALPHA, BETA)
    eps = np.sqrt(-c2)
    theta = np.arccos(c1 / (2 * eps))
    l = oracle.lib()
    it = l.or_integrator_build(RHO, oracle._dp(lam), 4, ALPHA, BETA, H, -1)
    ones = np.ones(4)
    q = [np.ctypeslib.as_array(l.or_integrator_step(it, oracle._dp(ones)), shape=(4,)).copy()]
    for _ in range(600):
        q.append(np.ctypeslib.as_array(l.or_integrator_step_free(it), shape=(4,)).copy())
    l.or_integrator_free(it)
    q = np.array(q)
    k = np.arange(601)[:, None]
    ref = c3 * eps ** k * np.sin((k + 1) * theta) / np.sin(theta)
    np.testing.assert_allclose(q, ref, rtol=0, atol=2e-9 * np.abs(ref).max())


def test_overdamped_mode_is_nan(oracle):
    # b < a^2/4 -> NaN coefficients, no guard in the reference
    c1, c2, c3 = oracle.iir_coeffs(np.array([RHO * 1.0]), RHO, 10.0, 0.0)
    assert np.isnan(c1[0]) and np.isnan(c3[0])
