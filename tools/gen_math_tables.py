#!/usr/bin/env python3
"""Derives the polynomial coefficients of csrc/smpc_math.hpp (exp / atan / sin / cos kernels) with a Remez exchange
in 60-digit arithmetic (mpmath) and prints them as C initialisers. Run once; the output is pasted into
smpc_math.hpp (kMathTable). Nothing here runs at build or test time except tests/test_math.py, which re-derives
nothing: it checks the shipped table against libm on the host.

    python tools/gen_math_tables.py
"""
import mpmath as mp

mp.mp.dps = 60


def remez(f, powers, lo, hi, weight, iters=40, grid=4000):
    """Minimax fit of sum_j c_j x^powers[j] to f on [lo, hi] for the error weight(x) * (f - p)."""
    n = len(powers)
    lo, hi = mp.mpf(lo), mp.mpf(hi)
    # initial reference: Chebyshev extrema
    xs = [(lo + hi) / 2 - (hi - lo) / 2 * mp.cos(mp.pi * k / n) for k in range(n + 1)]
    coef = None
    for _ in range(iters):
        A = mp.matrix(n + 1, n + 1)
        b = mp.matrix(n + 1, 1)
        for i, x in enumerate(xs):
            for j, pw in enumerate(powers):
                A[i, j] = x ** pw
            A[i, n] = (-1) ** i / weight(x)
            b[i] = f(x)
        sol = mp.lu_solve(A, b)
        coef = [sol[j] for j in range(n)]
        E = abs(sol[n])

        def err(x):
            return weight(x) * (f(x) - sum(c * x ** pw for c, pw in zip(coef, powers)))

        # locate extrema on a Chebyshev-spaced grid, refine by golden section
        g = [(lo + hi) / 2 - (hi - lo) / 2 * mp.cos(mp.pi * k / grid) for k in range(grid + 1)]
        ev = [err(x) for x in g]
        cand = []
        for k in range(grid + 1):
            l = ev[k - 1] if k > 0 else None
            r = ev[k + 1] if k < grid else None
            a = abs(ev[k])
            if (l is None or a >= abs(l)) and (r is None or a >= abs(r)):
                cand.append(k)
        # keep alternating-sign extrema, largest of each run
        runs = []
        for k in cand:
            s = 1 if ev[k] > 0 else -1
            if runs and runs[-1][0] == s:
                if abs(ev[k]) > abs(ev[runs[-1][1]]):
                    runs[-1] = (s, k)
            else:
                runs.append((s, k))
        while len(runs) > n + 1:  # drop the smaller end
            if abs(ev[runs[0][1]]) < abs(ev[runs[-1][1]]):
                runs.pop(0)
            else:
                runs.pop()
        if len(runs) < n + 1:
            break
        new_xs = [g[k] for _, k in runs]
        emax = max(abs(ev[k]) for _, k in runs)
        xs = new_xs
        if emax - E < E * mp.mpf("1e-6"):
            break
    return coef, emax


def show(name, coef, emax):
    print(f"// {name}: max weighted error {mp.nstr(emax, 4)}")
    for c in coef:
        print(f"  {float(c)!r},")


if __name__ == "__main__":
    ln2 = mp.log(2)
    # exp(r), |r| <= ln2/2 (+ slack): free fit of degree 11 / 12; c0 and c1 come out as 1 to 1e-18 and are taken as 1
    for deg in (11, 12):
        c, e = remez(mp.exp, list(range(deg + 1)), -ln2 / 2 * mp.mpf("1.01"), ln2 / 2 * mp.mpf("1.01"),
                     lambda x: 1 / mp.exp(x))
        show(f"exp degree {deg} (relative error)", c, e)

    # atan(t) = t * G(s), s = t^2 in [0, 1], G(s) ~ g(s) = atan(sqrt s) / sqrt s; relative error
    def g(s):
        if s == 0:
            return mp.mpf(1)
        t = mp.sqrt(s)
        return mp.atan(t) / t
    for deg in (20, 21, 22):
        c, e = remez(g, list(range(deg + 1)), 0, 1, lambda s: 1 / g(s))
        show(f"atan: G degree {deg} in s = t^2 (relative error of atan)", c, e)

    # sin(r) = r * S(z), cos(r) = C(z), z = r^2, |r| <= pi/4 (+ slack)
    q = (mp.pi / 4 * mp.mpf("1.01")) ** 2

    def sn(z):
        if z == 0:
            return mp.mpf(1)
        r = mp.sqrt(z)
        return mp.sin(r) / r
    for deg in (6, 7):
        c, e = remez(sn, list(range(deg + 1)), 0, q, lambda z: 1 / sn(z))
        show(f"sin: S degree {deg} in z = r^2 (relative error of sin)", c, e)
        c, e = remez(lambda z: mp.cos(mp.sqrt(z)), list(range(deg + 1)), 0, q, lambda z: 1 / mp.cos(mp.sqrt(z)))
        show(f"cos: C degree {deg} in z = r^2 (relative error of cos)", c, e)

    # atan2 of a UNIT vector (smpc_math.hpp: atan2_unit): after the octant reduction the angle a in [0, pi/4] is known by
    # its sine mn and cosine mx. Nodes S_k = k / 32 (k = rint(32 mn) = 0..23), C_k = sqrt(1 - S_k^2), A_k = asin(S_k):
    # s' = mn C_k - mx S_k = sin(a - A_k), |s'| <= 0.0226, and a = A_k + asin(s'), asin(s') = s' * U(x), x = s'^2.
    xmax = mp.mpf("0.0232") ** 2

    def u(x):
        if x == 0:
            return mp.mpf(1)
        s = mp.sqrt(x)
        return mp.asin(s) / s
    for deg in (3, 4):
        c, e = remez(u, list(range(deg + 1)), 0, xmax, lambda x: 1 / u(x))
        show(f"asin: U degree {deg} in x = s'^2 (relative error of asin(s'))", c, e)
    print("// atan2_unit nodes: C_k, A_k for k = 0..23")
    for k in range(24):
        s = mp.mpf(k) / 32
        print(f"  {float(mp.sqrt(1 - s * s))!r}, {float(mp.asin(s))!r},")
