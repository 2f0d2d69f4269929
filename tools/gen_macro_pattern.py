#!/usr/bin/env python
"""Generates topopt_in_petsc_amd/csrc/macro_pattern.h: the structural non-zero pattern of the level-1 Galerkin
element operator written in the Walsh-Hadamard basis of BOTH the 8 coarse corners and the 8 child moduli,

    K_E(E_children) = sum_c E_c W_c^T KE W_c  =  H^T [ sum_sigma ehat_sigma G_sigma ] H,
    ehat_sigma = sum_c (-1)^{|sigma & c|} E_c,    G_sigma = H (1/8 sum_c (-1)^{|sigma & c|} W_c^T KE W_c) H^T / 64.

The pattern only depends on the reflection symmetries of a box element; it is taken from two
anisotropic boxes (identical; a cube only adds accidental zeros).  Values are computed at run time (matfree_tile.h: make_macro_tensor)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hex8_ke_box(nu, hx, hy, hz):
    """24x24 stiffness of a box hex8 element (E = 1, 2x2x2 Gauss), reference corner order; only its symmetry
    structure matters here"""
    lam, mu = nu / ((1 + nu) * (1 - 2 * nu)), 1 / (2 * (1 + nu))
    Cm = np.zeros((6, 6))
    Cm[:3, :3] = lam
    Cm[np.arange(3), np.arange(3)] += 2 * mu
    Cm[np.arange(3, 6), np.arange(3, 6)] = mu
    xi = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], float)
    KE = np.zeros((24, 24))
    gp = 1 / np.sqrt(3)
    for a in (-gp, gp):
        for b in (-gp, gp):
            for c in (-gp, gp):
                p = np.array([a, b, c])
                dN = np.zeros((3, 8))
                for n in range(8):
                    for d in range(3):
                        f = 0.125 * xi[n, d]
                        for o in range(3):
                            if o != d:
                                f *= 1 + xi[n, o] * p[o]
                        dN[d, n] = f
                dN = dN / (np.array([hx, hy, hz])[:, None] / 2)
                B = np.zeros((6, 24))
                for n in range(8):
                    B[0, 3 * n], B[1, 3 * n + 1], B[2, 3 * n + 2] = dN[0, n], dN[1, n], dN[2, n]
                    B[3, 3 * n], B[3, 3 * n + 1] = dN[1, n], dN[0, n]
                    B[4, 3 * n + 1], B[4, 3 * n + 2] = dN[2, n], dN[1, n]
                    B[5, 3 * n], B[5, 3 * n + 2] = dN[2, n], dN[0, n]
                KE += B.T @ Cm @ B * (hx * hy * hz / 8)
    return KE

M2A = [0, 1, 3, 2, 4, 5, 7, 6]


def tensors(nu, hx, hy, hz):
    KE = hex8_ke_box(nu, hx, hy, hz)
    P = np.zeros((24, 24))
    for m in range(8):
        for c in range(3):
            P[3 * M2A[m] + c, 3 * m + c] = 1
    Kn = P.T @ KE @ P

    def W(ch):
        Wm = np.zeros((8, 8))
        for m in range(8):
            p = [(ch >> a & 1) + (m >> a & 1) for a in range(3)]
            for M in range(8):
                w = 1.0
                for a in range(3):
                    w *= p[a] / 2 if (M >> a) & 1 else 1 - p[a] / 2
                Wm[m, M] = w
        return np.kron(Wm, np.eye(3))
    Mc = [W(c).T @ Kn @ W(c) for c in range(8)]
    H2 = np.array([[1, 1], [1, -1]])
    H = np.kron(np.kron(H2, np.kron(H2, H2)), np.eye(3))
    G = []
    for s in range(8):
        Ns = sum((-1) ** bin(s & c).count("1") * Mc[c] for c in range(8)) / 8
        G.append(H @ Ns @ H.T / 64)
    return Mc, H, G


def main():
    pats = []
    for geo in ((0.3, 1, 1, 1), (0.3, 1.0, 0.7, 1.3), (0.2, 0.5, 2.0, 1.1)):
        Mc, H, G = tensors(*geo)
        scale = max(abs(g).max() for g in G)
        pats.append({(s, i, j) for s in range(8) for i in range(24) for j in range(24) if abs(G[s][i, j]) > 1e-12 * scale})
        # self-check of the identity on random data
        rng = np.random.default_rng(0)
        E, u = rng.random(8), rng.standard_normal(24)
        y0 = sum(E[c] * Mc[c] for c in range(8)) @ u
        eh = [sum((-1) ** bin(s & c).count("1") * E[c] for c in range(8)) for s in range(8)]
        y1 = H.T @ (sum(eh[s] * G[s] for s in range(8)) @ (H @ u))
        assert np.allclose(y0, y1, rtol=1e-12, atol=1e-12 * abs(y0).max())
    # a cube has a few accidental zeros on top of the structural ones
    assert pats[1] == pats[2] and pats[0] <= pats[1], [len(p) for p in pats]
    full = sorted(pats[1])
    assert all((sg, j, i) in pats[1] for sg, i, j in full)          # every G_sigma is symmetric
    mem = [t for t in full if t[1] <= t[2]]                        # upper triangle
    # entries that are equal up to sign for EVERY box (checked on the two anisotropic ones) share one constant
    Gs = [tensors(*geo)[2] for geo in ((0.3, 1.0, 0.7, 1.3), (0.2, 0.5, 2.0, 1.1))]
    cls, reps = [], []
    for (sg, i, j) in mem:
        v = [G[sg][i, j] for G in Gs]
        for c, (rs, ri, rj) in enumerate(reps):
            r = [G[rs][ri, rj] for G in Gs]
            for sign in (1, -1):
                if all(abs(a - sign * b) <= 1e-11 * max(abs(a), abs(b)) for a, b in zip(v, r)):
                    cls.append((c, sign))
                    break
            else:
                continue
            break
        else:
            reps.append((sg, i, j))
            cls.append((len(reps) - 1, 1))
    out = os.path.join(ROOT, "topopt_in_petsc_amd", "csrc", "macro_pattern.h")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_macro_pattern.py -- do not edit.\n"
                "// Level-1 Galerkin element operator in the Walsh-Hadamard basis of the coarse corners and of the 8 child\n"
                "// moduli: the matrices G_sigma are symmetric; MACG_M_* lists their structurally non-zero upper-triangle\n"
                "// entries (sigma, row = 3*mode + component, column >= row), each equal to +-G[class] for any box element\n"
                "// (MACG_M_CLS / MACG_M_SGN); class c is represented by entry MACG_REP[c] of that list.\n#pragma once\n")
        f.write("constexpr int MACG_M = %d;   // entries\nconstexpr int MACG_N = %d;   // distinct constants\n" % (len(mem), len(reps)))
        for name, vals in (("MACG_M_SIG", [t[0] for t in mem]), ("MACG_M_ROW", [t[1] for t in mem]),
                           ("MACG_M_COL", [t[2] for t in mem]), ("MACG_M_CLS", [c for c, sg in cls])):
            f.write("static const unsigned char %s[MACG_M] = {%s};\n" % (name, ", ".join(str(v) for v in vals)))
        f.write("static const signed char MACG_M_SGN[MACG_M] = {%s};\n" % ", ".join(str(sg) for c, sg in cls))
        f.write("static const unsigned char MACG_REP[MACG_N] = {%s};\n" % ", ".join(str(mem.index(r)) for r in reps))
        f.write("\n// fhat[c][p] = sum_sigma eh[sigma] * (G_sigma uhat)[3p + c];  G = the MACG_N constants (scalar loads)\n"
                "__device__ __forceinline__ void macg_apply(const double *__restrict__ G, const double (&u)[3][8],\n"
                "                                           const double (&eh)[8], double (&f)[3][8]) {\n    double g;\n")
        seen = set()

        def acc(row, col, sign):
            ff = "f[%d][%d]" % (row % 3, row // 3)
            uu = "u[%d][%d]" % (col % 3, col // 3)
            gg = "g" if sign > 0 else "-g"
            line = "    %s = %s;\n" % (ff, ("%s * %s" % (gg, uu)) if row not in seen else "fma(%s, %s, %s)" % (gg, uu, ff))
            seen.add(row)
            return line
        # constants in class order so that the scalar loads stream through the table once
        order = sorted(range(len(mem)), key=lambda k: (cls[k][0], k))
        for k in order:
            sg, i, j = mem[k]
            c, sign = cls[k]
            f.write("    g = eh[%d] * G[%d];\n" % (sg, c))
            f.write(acc(i, j, sign))
            if i != j:
                f.write(acc(j, i, sign))
        for row in range(24):
            if row not in seen:
                f.write("    f[%d][%d] = 0.0;\n" % (row % 3, row // 3))
        f.write("}\n")
    pat = reps
    print("wrote", out, len(reps), "constants,", len(mem), "upper-triangle entries of", len(full))


if __name__ == "__main__":
    main()
