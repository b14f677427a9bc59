#!/usr/bin/env python3
"""The identities the a != 0 closed-form RHS (accel_spin_true, raytracegr.jl_amd/csrc/rtgr_physics.hpp) rests on, checked
in 40-digit arithmetic (mpmath) at random points, and the regrouped acceleration itself against the general Kerr–Schild-form
contraction (ksform_accel's formula) and against the Christoffel contraction of the 4x4 metric (src/RayTraceGR.jl:321-331,
:358-370 — the definition).

With q = rho² − a², Σ = sqrt(q² + 4a²z²), r² = (q + Σ)/2 (the textbook Kerr–Schild radius), w = 1/(r² + a²),
k = ((r x + a y) w, (r y − a x) w, z/r), f = 2M r³/(r⁴ + a²z²):

    (1) r⁴ + a²z² = r² Σ                      (2) |k|² = 1                 (3) k^j ∂_j k_i = 0
    (4) k^i ∂_d k_i = 0                        (5) k·∇r = 1
    (6) ∇r = (r/Σ)(x,y,z) + (a²z/(rΣ)) ẑ      (7) f = 2M r/Σ
    (8) ∂f/∂r|_z = f ψ,  ψ = 3/r − 4r/Σ        (9) ∂f/∂z|_r = f φ,  φ = −2a²z/(r²Σ)
    (10) the regrouped u̇ of accel_spin_true == −Γ^a_bc u^b u^c of g = η + f k⊗k

    python tools/check_identities.py [--points N] [--digits D]     exit code 0 iff every residual < 10^-(D-8)
"""
import argparse
import random
import sys

import mpmath as mp


def fields(x, y, z, M, a):
    q = x * x + y * y + z * z - a * a
    sig = mp.sqrt(q * q + 4 * a * a * z * z)
    r2 = (q + sig) / 2
    r = mp.sqrt(r2)
    w = 1 / (r2 + a * a)
    k = [(r * x + a * y) * w, (r * y - a * x) * w, z / r]
    f = 2 * M * r ** 3 / (r ** 4 + a * a * z * z)
    return q, sig, r, w, k, f


def grad(fun, p, h=None):
    """central differences in 40+ digit arithmetic are exact enough; mp.diff does Richardson extrapolation"""
    return [mp.diff(lambda t, i=i: fun(*[p[j] if j != i else t for j in range(3)]), p[i]) for i in range(3)]


def metric(xx, M, a):
    _, _, _, _, k, f = fields(xx[1], xx[2], xx[3], M, a)
    k4 = [mp.mpf(1)] + k
    eta = [-1, 1, 1, 1]
    return [[(eta[i] if i == j else 0) + f * k4[i] * k4[j] for j in range(4)] for i in range(4)]


def christoffel_accel(xx, u, M, a):
    g = mp.matrix(metric(xx, M, a))
    gi = g ** -1
    dg = [[[mp.mpf(0)] * 4 for _ in range(4)] for _ in range(4)]       # dg[a][b][c] = ∂_c g_ab (stationary: c = 0 is zero)
    for c in range(1, 4):
        for i in range(4):
            for j in range(4):
                dg[i][j][c] = mp.diff(lambda t: metric([xx[m] if m != c else t for m in range(4)], M, a)[i][j], xx[c])
    ud = []
    for p in range(4):
        acc = mp.mpf(0)
        for b in range(4):
            for c in range(4):
                G = sum(gi[p, d] * (dg[d][b][c] + dg[d][c][b] - dg[b][c][d]) for d in range(4)) / 2
                acc -= G * u[b] * u[c]
        ud.append(acc)
    return ud


def accel_spin_true(xs, u, M, a):
    """accel_spin_true of rtgr_physics.hpp, operation for operation (without the one-rsq trick, which is algebra-neutral)"""
    x, y, z = xs
    ut, ux, uy, uz = u
    a2 = a * a
    q = x * x + y * y + z * z - a2
    a2z = a2 * z
    sig = mp.sqrt(4 * a2z * z + q * q)
    is_ = 1 / sig
    r2 = (q + sig) / 2
    r = mp.sqrt(r2)
    ir = 1 / r
    w = 1 / (r2 + a2)
    rid = r * is_
    rz = a2z * (is_ * ir)
    phi = -(rz * ir) * 2
    psi = -4 * rid + 3 * ir
    rw, aw = r * w, a * w
    k0, k1, k2 = rw * x + aw * y, rw * y - aw * x, z * ir
    xu2 = x * ux + y * uy
    xu = z * uz + xu2
    Ku2 = k0 * ux + k1 * uy
    K = ut + k2 * uz + Ku2
    D = rid * xu + rz * uz
    rw2m = -2 * rw
    iruz = ir * uz
    kru = w * xu2 + rw2m * Ku2 - k2 * iruz
    A = kru * D + rw * (ux * ux + uy * uy) + iruz * uz
    inner = psi * D + phi * uz
    f = 2 * M * rid
    Kf = f * K
    hf = (K / 2) * Kf
    udt = hf * f * (phi * k2 + psi) + Kf * inner + f * A
    E = Kf * D
    cr = Kf * kru + hf * psi
    nck = rw2m * E + udt
    crr = cr * rid
    cx = -E * w + crr
    rot = (2 * a * w) * Kf
    return [udt, -k0 * nck + x * cx - rot * uy, -k1 * nck + y * cx + rot * ux,
            z * (ir * (E * ir - udt) + crr) + cr * rz + hf * phi]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=6)
    ap.add_argument("--digits", type=int, default=40)
    args = ap.parse_args()
    mp.mp.dps = args.digits + 10
    tol = mp.mpf(10) ** (-(args.digits - 8))
    rnd = random.Random(20261004)
    worst = {}

    def note(name, res):
        worst[name] = max(worst.get(name, mp.mpf(0)), abs(res))

    for _ in range(args.points):
        M = mp.mpf(1)
        a = mp.mpf(rnd.choice(["0.8", "0.998", "0.3"]))
        p = [mp.mpf(rnd.uniform(-6, 6)) for _ in range(3)]
        if sum(c * c for c in p) < 4:
            p[0] += 4
        x, y, z = p
        q, sig, r, w, k, f = fields(x, y, z, M, a)
        note("(1) r^4 + a^2 z^2 = r^2 Sigma", (r ** 4 + a * a * z * z) / (r * r * sig) - 1)
        note("(2) |k|^2 = 1", sum(c * c for c in k) - 1)
        dk = [grad(lambda X, Y, Z, i=i: fields(X, Y, Z, M, a)[4][i], p) for i in range(3)]   # dk[i][j] = ∂_j k_i
        for i in range(3):
            note("(3) k^j d_j k_i = 0", sum(k[j] * dk[i][j] for j in range(3)))
            note("(4) k^i d_d k_i = 0", sum(k[j] * dk[j][i] for j in range(3)))
        dr = grad(lambda X, Y, Z: fields(X, Y, Z, M, a)[2], p)
        note("(5) k.grad r = 1", sum(k[i] * dr[i] for i in range(3)) - 1)
        want = [r / sig * x, r / sig * y, r / sig * z + a * a * z / (r * sig)]
        for i in range(3):
            note("(6) grad r", dr[i] - want[i])
        note("(7) f = 2M r/Sigma", f / (2 * M * r / sig) - 1)
        # f as a function of (r, z): f = 2M r^3/(r^4 + a^2 z^2)
        fr = mp.diff(lambda R: 2 * M * R ** 3 / (R ** 4 + a * a * z * z), r)
        fz = mp.diff(lambda Z: 2 * M * r ** 3 / (r ** 4 + a * a * Z * Z), z)
        note("(8) df/dr = f psi", fr / f - (3 / r - 4 * r / sig))
        note("(9) df/dz = f phi", fz / f - (-2 * a * a * z / (r * r * sig)))
        u = [mp.mpf(rnd.uniform(-1, 1)) for _ in range(4)]
        got = accel_spin_true(p, u, M, a)
        ref = christoffel_accel([mp.mpf(0)] + p, u, M, a)
        scale = max(abs(c) for c in ref) + mp.mpf("1e-30")
        for i in range(4):
            note("(10) regrouped accel = -Gamma u u", (got[i] - ref[i]) / scale)
    bad = 0
    for name, res in worst.items():
        ok = res < tol
        bad += not ok
        print(f"{'ok  ' if ok else 'FAIL'} {name:38s} max residual {mp.nstr(res, 3)}")
    print(f"{args.points} points, {args.digits} digits, tolerance {mp.nstr(tol, 2)}: {'all identities hold' if not bad else str(bad) + ' FAILED'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
