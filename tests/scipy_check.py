"""Independent scipy.sparse re-derivation of the operators, used to cross-check
the C oracle (third opinion; test infrastructure)."""
import numpy as np
import scipy.sparse as sp

LOC = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]])


def elem_nodes(ex, ey, ez):
    nx, ny = ex + 1, ey + 1
    k, j, i = np.meshgrid(np.arange(ez), np.arange(ey), np.arange(ex), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()  # x fastest
    return np.stack([(i + l[0]) + nx * ((j + l[1]) + ny * (k + l[2])) for l in LOC], axis=1)


def assemble(ex, ey, ez, KE, E=None, N=None, dof=3):
    nodes = elem_nodes(ex, ey, ez)
    nel = nodes.shape[0]
    ed = 8 * dof
    edof = (dof * nodes[:, :, None] + np.arange(dof)[None, None, :]).reshape(nel, ed)
    rows = np.repeat(edof, ed, axis=1).ravel()
    cols = np.tile(edof, (1, ed)).ravel()
    E = np.ones(nel) if E is None else E
    vals = (E[:, None] * np.asarray(KE).ravel()[None, :]).ravel()
    n = dof * (ex + 1) * (ey + 1) * (ez + 1)
    K = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    if N is not None:
        D = sp.diags(N)
        K = (D @ K @ D + sp.diags(1 - N)).tocsr()
    return K


def interp1d(nc):
    nf = 2 * (nc - 1) + 1
    P = sp.lil_matrix((nf, nc))
    for i in range(nf):
        if i % 2 == 0:
            P[i, i // 2] = 1
        else:
            P[i, i // 2] = 0.5
            P[i, i // 2 + 1] = 0.5
    return P.tocsr()


def interp3d(ncx, ncy, ncz, dof):
    P = sp.kron(interp1d(ncz), sp.kron(interp1d(ncy), interp1d(ncx)))
    return sp.kron(P, sp.identity(dof)).tocsr()


def filter_matrix(ex, ey, ez, h, R):
    """H_ij = R - dist if dist < R  (Filter.cc:417-433), dense-offset construction."""
    conn = int(max(np.ceil(R / h) - 1, 0))
    conn = min(conn, ex // 2, ey // 2, ez // 2)
    k, j, i = np.meshgrid(np.arange(ez), np.arange(ey), np.arange(ex), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    rows, cols, vals = [], [], []
    for dk in range(-conn, conn + 1):
        for dj in range(-conn, conn + 1):
            for di in range(-conn, conn + 1):
                d = h * np.sqrt(di * di + dj * dj + dk * dk)
                if d >= R:
                    continue
                ok = (i + di >= 0) & (i + di < ex) & (j + dj >= 0) & (j + dj < ey) & (k + dk >= 0) & (k + dk < ez)
                r = (i + ex * (j + ey * k))[ok]
                c = ((i + di) + ex * ((j + dj) + ey * (k + dk)))[ok]
                rows.append(r)
                cols.append(c)
                vals.append(np.full(r.size, R - d))
    n = ex * ey * ez
    return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()


def elem_to_node_T(ex, ey, ez):
    nodes = elem_nodes(ex, ey, ez)
    nel = nodes.shape[0]
    rows = nodes.ravel()
    cols = np.repeat(np.arange(nel), 8)
    return sp.coo_matrix((np.full(rows.size, 0.125), (rows, cols)),
                         shape=((ex + 1) * (ey + 1) * (ez + 1), nel)).tocsr()
