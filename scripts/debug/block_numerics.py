"""Numerics prototype of the block state-space oscillator bank (K1b), fp32-emulated in numpy,
against an fp64 per-sample reference.  Per buffer: sample 0 literal (velocity form), then 32 blocks
of 16 samples: staged state x_n, output y[1+16n+j] = sum_m a_{j+1}[m] Q_m + b_{j+1}[m] D_m (MFMA =
k-ordered fmaf chain), coarse step x <- P x with P = A^16 in the (q, d) basis."""
import sys
import numpy as np
sys.path.insert(0, ".")
from openpbso_amd import synth

f32 = np.float32
def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)

def coeffs(lam, rho, alpha, beta, h=1 / 44100.0):
    om0 = np.sqrt(lam / rho); xi = 0.5 * (alpha / om0 + beta * om0); a = 2 * xi * om0; b = om0 ** 2
    eps = np.exp(-a / 2 * h); th = h * np.sqrt(b - a * a / 4); gam = np.arcsin(a / (2 * np.sqrt(b)))
    om = np.sqrt(b); omd = np.sqrt(b - a ** 2 / 4)
    c1 = 2 * eps * np.cos(th); c2 = -eps ** 2
    c3 = 2 * (eps * np.cos(th + gam) - eps ** 2 * np.cos(2 * th + gam)) / (3 * om * omd) * 1e9
    return c1, c2, c3

def run(M=512, NB=86, J=16, seed=1, f_lo=100.0):
    B = 513
    lam = synth.eigenvalues(M, seed, f_lo=f_lo)
    c1, c2, c3 = coeffs(lam, synth.RHO, synth.ALPHA, synth.BETA)
    rng = np.random.default_rng(seed)
    t = np.abs(rng.standard_normal(M)) * 1e7 + 1e5              # transfer weights
    hits = rng.random(NB) < 0.233
    S = rng.standard_normal((NB, M)) * 1e-3
    # ---- fp64 reference (direct form, modal_integrator.h:109-110) ----
    q1 = np.zeros(M); q2 = np.zeros(M); yref = np.zeros(NB * B)
    for b in range(NB):
        for k in range(B):
            f = c3 * S[b] if (hits[b] and k == 0) else 0.0
            q = c1 * q1 + c2 * q2 + f
            q2 = q1; q1 = q
            yref[b * B + k] = t @ q
    # ---- block form, fp64 coefficient build in the (q, d) basis: x = (q_k, d_k = q_k - q_{k-1}) ----
    # one step: d' = -c2 d - e q, q' = q + d'  (e = 1 - c1 - c2)   =>  A = [[1 - e, -c2], [-e, -c2]]
    e = 1 - c1 - c2
    A = np.zeros((M, 2, 2)); A[:, 0, 0] = 1 - e; A[:, 0, 1] = -c2; A[:, 1, 0] = -e; A[:, 1, 1] = -c2
    Ap = np.tile(np.eye(2), (M, 1, 1)); W = np.zeros((J, M, 2))
    for j in range(J):
        Ap = A @ Ap                                              # A^(j+1)
        W[j] = Ap[:, 0, :]
    P = Ap                                                       # A^J
    Wf = W.astype(f32)
    e11 = (P[:, 0, 0] - 1).astype(f32); p12 = P[:, 0, 1].astype(f32); p21 = P[:, 1, 0].astype(f32); p22 = P[:, 1, 1].astype(f32)
    ca = (-c2).astype(f32); cb = (-e).astype(f32); g = c3.astype(f32); tf = t.astype(f32)
    Q = np.zeros(M, f32); D = np.zeros(M, f32); y = np.zeros(NB * B, f32)
    NBLK = (B - 1) // J
    assert NBLK * J == B - 1
    for b in range(NB):
        # sample 0 literal, scaled state
        a_ = ca * D
        a_ = fma(cb, Q, a_)
        if hits[b]:
            a_ = fma((S[b].astype(f32) * g) * tf, np.ones(M, f32), a_)     # g_scaled = t g S (one rounding each)
        D = a_; Q = Q + a_
        y[b * B] = np.sum(Q.astype(np.float64)).astype(f32)       # wave reduction (order differs on device; fine)
        X = np.zeros((NBLK, M, 2), f32)
        for n in range(NBLK):
            X[n, :, 0] = Q; X[n, :, 1] = D
            qa = fma(e11, Q, Q); qn = fma(p12, D, qa)
            da = p21 * Q; dn = fma(p22, D, da)
            Q, D = qn, dn
        # MFMA: k-ordered fmaf chain over (mode, comp)
        acc = np.zeros((J, NBLK), f32)
        for m in range(M):
            for c in range(2):
                acc = fma(Wf[:, m, c][:, None], X[:, m, c][None, :], acc)
        y[b * B + 1:(b + 1) * B] = acc.T.reshape(-1)
    err = np.abs(y - yref).max() / np.abs(yref).max()
    l2 = np.linalg.norm(y - yref) / np.linalg.norm(yref)
    return err, l2

if __name__ == "__main__":
    for args in [dict(M=512, J=16), dict(M=512, J=16, f_lo=20.0), dict(M=512, J=32), dict(M=128, J=16, seed=3)]:
        err, l2 = run(**args)
        print(args, f"max|err|/peak = {err:.3e}  rel L2 = {l2:.3e}")


def bf16_trunc(x):
    """top 16 bits of the fp32 pattern (truncation), returned as fp32"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return (u & np.uint32(0xFFFF0000)).view(np.float32)


def bf16_round_half_up(x):
    """(bits + 0x8000) & 0xFFFF0000: what the kernel's integer split does for the hi part of a state"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x8000)) & np.uint32(0xFFFF0000)).view(np.float32)


def run_bf16x3(M=512, NB=86, seed=1, f_lo=100.0, terms=3):
    """same block form, but the output projection runs as bf16 x bf16 -> fp32 products of the split
    operands: hi.hi + hi.lo + lo.hi (terms = 3), state recurrence unchanged in fp32"""
    B, J = 513, 16
    lam = synth.eigenvalues(M, seed, f_lo=f_lo)
    c1, c2, c3 = coeffs(lam, synth.RHO, synth.ALPHA, synth.BETA)
    rng = np.random.default_rng(seed)
    t = np.abs(rng.standard_normal(M)) * 1e7 + 1e5
    hits = rng.random(NB) < 0.233
    S = rng.standard_normal((NB, M)) * 1e-3
    q1 = np.zeros(M); q2 = np.zeros(M); yref = np.zeros(NB * B)
    for b in range(NB):
        for k in range(B):
            f = c3 * S[b] if (hits[b] and k == 0) else 0.0
            q = c1 * q1 + c2 * q2 + f
            q2 = q1; q1 = q
            yref[b * B + k] = t @ q
    e = 1 - c1 - c2
    A = np.zeros((M, 2, 2)); A[:, 0, 0] = 1 - e; A[:, 0, 1] = -c2; A[:, 1, 0] = -e; A[:, 1, 1] = -c2
    Ap = np.tile(np.eye(2), (M, 1, 1)); W = np.zeros((J, M, 2))
    for j in range(J):
        Ap = A @ Ap
        W[j] = Ap[:, 0, :]
    P = Ap
    Wf = W.astype(f32).reshape(J, 2 * M)
    Wf = (Wf.astype(np.float64) * (1.0 + 7.2e-6)).astype(f32)     # TRUNC_SPLIT_GAIN (kernels.h)
    Whi = bf16_trunc(Wf); Wlo = bf16_trunc(Wf - Whi)
    e11 = (P[:, 0, 0] - 1).astype(f32); p12 = P[:, 0, 1].astype(f32); p21 = P[:, 1, 0].astype(f32); p22 = P[:, 1, 1].astype(f32)
    ca = (-c2).astype(f32); cb = (-e).astype(f32); g = c3.astype(f32); tf = t.astype(f32)
    Q = np.zeros(M, f32); D = np.zeros(M, f32); y = np.zeros(NB * B, f32)
    NBLK = (B - 1) // J
    for b in range(NB):
        a_ = ca * D
        a_ = fma(cb, Q, a_)
        if hits[b]:
            a_ = fma((S[b].astype(f32) * g) * tf, np.ones(M, f32), a_)
        D = a_; Q = Q + a_
        y[b * B] = np.sum(Q.astype(np.float64)).astype(f32)
        X = np.zeros((NBLK, 2 * M), f32)
        for n in range(NBLK):
            X[n, 0::2] = Q; X[n, 1::2] = D
            qa = fma(e11, Q, Q); qn = fma(p12, D, qa)
            da = p21 * Q; dn = fma(p22, D, da)
            Q, D = qn, dn
        Xhi = bf16_trunc(X); Xlo = bf16_trunc(X - Xhi)        # the kernel: both parts truncated, the mean loss folded into W
        # products of bf16 operands are exact in fp32; accumulate in fp32 (chunks of 32 in fp64 ~ the MFMA's internal sum)
        acc = np.zeros((J, NBLK), np.float64)
        pairs = [(Whi, Xhi), (Whi, Xlo), (Wlo, Xhi)][:terms]
        for (a, x) in pairs:
            acc += (a.astype(np.float64) @ x.astype(np.float64).T)
        y[b * B + 1:(b + 1) * B] = acc.T.reshape(-1).astype(f32)
    err = np.abs(y - yref).max() / np.abs(yref).max()
    l2 = np.linalg.norm(y - yref) / np.linalg.norm(yref)
    return err, l2


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "bf16":
    for args in [dict(terms=3), dict(terms=1), dict(terms=3, f_lo=20.0), dict(terms=3, M=128, seed=3)]:
        err, l2 = run_bf16x3(**args)
        print("bf16 split", args, f"max|err|/peak = {err:.3e}  rel L2 = {l2:.3e}")
