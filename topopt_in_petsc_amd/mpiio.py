"""Result and restart files around the hot path.  Host-side I/O only.

`MPIIO` writes the reference's `output_00000.dat` container (MPIIO.cc:207-377 header, :380-714 data,
WriteVTK :147-205) so that the reference's own converter `bin2vtu_v3.py` turns it into ParaView files:

    "TopOpt result version 1.1\\n\\x01"
    uint64 nDom | nPointsT[nDom] | nCellsT[nDom] | nPFields[nDom] | nCFields[nDom] | nodesPerElement
    "<point field names>\\x01<cell field names>\\x01"
    float32 points (x,y,z per point; every rank's GHOSTED local node set, rank after rank)
    uint64 connectivity (8 per cell, shifted by the points written by lower ranks) | offsets | types (12)
    per dump: uint64 iteration | point fields (field-major, float32) | cell fields (field-major, float32)

Multi-rank: sections are appended rank after rank (single node, shared file system).

Restart files: the PETSc binary Vec layout (big-endian int32 class id 1211214, int32 n, n float64) and the
ASCII "itr fscale" companion (TopOpt.cc:514-570; LinearElasticity.cc:447-478).  The class id / endianness are
from the PETSc documentation as recalled in SURVEY.md; they have not been checked against a PETSc installation.
"""
import os
import struct

import numpy as np

VEC_FILE_CLASSID = 1211214


def _barrier(nranks):
    if nranks > 1:
        import torch.distributed as dist
        dist.barrier()


class MPIIO:
    def __init__(self, part, h, pnames="ux, uy, uz", cnames="x, xTilde, xPhys", nPf=3, nCf=3,
                 filename="output_00000.dat", xc0=(0.0, 0.0, 0.0)):
        self.part, self.filename, self.nPf, self.nCf = part, filename, nPf, nCf
        p = part
        hx, hy, hz = (h, h, h) if np.isscalar(h) else h
        # ghosted local node set of this rank (local planes incl. ghosts), DMDA local ordering
        k, j, i = np.meshgrid(np.arange(p.nz_local), np.arange(p.ny), np.arange(p.nx), indexing="ij")
        pts = np.stack([xc0[0] + i.ravel() * hx, xc0[1] + j.ravel() * hy, xc0[2] + (k.ravel() + p.node_z0) * hz], axis=1)
        self.points = pts.astype(np.float32)
        self.nP = self.points.shape[0]
        # own elements in local node numbering, reference corner order (LinearElasticity.cc:819-826)
        ke, je, ie = np.meshgrid(np.arange(p.ez_own), np.arange(p.ey), np.arange(p.ex), indexing="ij")
        ie, je, ke = ie.ravel(), je.ravel(), ke.ravel()
        loc = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
        self.conn = np.stack([(ie + a) + p.nx * ((je + b) + p.ny * (ke + c)) for a, b, c in loc], axis=1).astype(np.uint64)
        self.nC = self.conn.shape[0]
        self.counts = self._allgather_counts()
        self._write_mesh(pnames, cnames)

    def _allgather_counts(self):
        if self.part.nranks == 1:
            return [(self.nP, self.nC)]
        import torch.distributed as dist
        out = [None] * self.part.nranks
        dist.all_gather_object(out, (self.nP, self.nC))
        return out

    def _append(self, writer):
        """rank-ordered append of one file section"""
        for r in range(self.part.nranks):
            if r == self.part.rank:
                with open(self.filename, "ab") as f:
                    writer(f)
            _barrier(self.part.nranks)

    def _write_mesh(self, pnames, cnames):
        p = self.part
        if p.rank == 0:
            with open(self.filename, "wb") as f:
                f.write(b"TopOpt result version 1.1\n\x01")
                nPT, nCT = sum(c[0] for c in self.counts), sum(c[1] for c in self.counts)
                f.write(struct.pack("<6Q", 1, nPT, nCT, self.nPf, self.nCf, 8))
                f.write(pnames.encode() + b"\x01" + cnames.encode() + b"\x01")
        _barrier(p.nranks)
        shift = sum(c[0] for c in self.counts[: p.rank])
        cells_before = sum(c[1] for c in self.counts[: p.rank])
        self._append(lambda f: f.write(self.points.tobytes()))
        self._append(lambda f: f.write((self.conn + np.uint64(shift)).tobytes()))
        offs = (np.arange(1, self.nC + 1, dtype=np.uint64) + np.uint64(cells_before)) * np.uint64(8)
        self._append(lambda f: f.write(offs.tobytes()))
        self._append(lambda f: f.write(np.full(self.nC, 12, dtype=np.uint64).tobytes()))

    def WriteVTK(self, U, x, xTilde, xPhys, itr):
        """MPIIO::WriteVTK (MPIIO.cc:147-205): U = local (ghosted) state, 3 per node; x.. = own elements"""
        to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        U = to_np(U).reshape(-1, 3)
        assert U.shape[0] == self.nP
        if self.part.rank == 0:
            with open(self.filename, "ab") as f:
                f.write(struct.pack("<Q", int(itr)))
        _barrier(self.part.nranks)
        for c in range(3):
            self._append(lambda f, c=c: f.write(U[:, c].astype(np.float32).tobytes()))
        for v in (x, xTilde, xPhys):
            self._append(lambda f, v=v: f.write(to_np(v).astype(np.float32).tobytes()))


def read_output(filename):
    """independent reader of the container (the layout bin2vtu_v3.py:23-104 expects)"""
    with open(filename, "rb") as f:
        raw = f.read()
    pos = raw.index(b"\x01") + 1
    info = raw[: pos - 1].decode()
    nDom, = struct.unpack_from("<Q", raw, pos)
    assert nDom == 1
    nPT, nCT, nPf, nCf, npe = struct.unpack_from("<5Q", raw, pos + 8)
    pos += 48
    e1 = raw.index(b"\x01", pos)
    e2 = raw.index(b"\x01", e1 + 1)
    pnames, cnames = raw[pos:e1].decode(), raw[e1 + 1:e2].decode()
    pos = e2 + 1
    pts = np.frombuffer(raw, dtype="<f4", count=3 * nPT, offset=pos).reshape(-1, 3)
    pos += 12 * nPT
    conn = np.frombuffer(raw, dtype="<u8", count=npe * nCT, offset=pos).reshape(-1, npe)
    pos += 8 * npe * nCT
    offs = np.frombuffer(raw, dtype="<u8", count=nCT, offset=pos)
    pos += 8 * nCT
    types = np.frombuffer(raw, dtype="<u8", count=nCT, offset=pos)
    pos += 8 * nCT
    dumps = []
    while pos < len(raw):
        it, = struct.unpack_from("<Q", raw, pos)
        pos += 8
        pf = np.frombuffer(raw, dtype="<f4", count=nPf * nPT, offset=pos).reshape(nPf, nPT)
        pos += 4 * nPf * nPT
        cf = np.frombuffer(raw, dtype="<f4", count=nCf * nCT, offset=pos).reshape(nCf, nCT)
        pos += 4 * nCf * nCT
        dumps.append((it, pf, cf))
    return dict(info=info, pnames=pnames, cnames=cnames, points=pts, conn=conn, offsets=offs, types=types, dumps=dumps)


# ---- restart files ---------------------------------------------------------------------------------
def write_petsc_vecs(filename, vecs):
    """VecView of several Vecs into one PETSc binary file (TopOpt.cc:558-564: x, xPhys, xo1, xo2, U, L)"""
    with open(filename, "wb") as f:
        for v in vecs:
            a = np.ascontiguousarray(v, dtype=np.float64)
            f.write(struct.pack(">ii", VEC_FILE_CLASSID, a.size))
            f.write(a.astype(">f8").tobytes())


def read_petsc_vecs(filename):
    out = []
    with open(filename, "rb") as f:
        raw = f.read()
    pos = 0
    while pos < len(raw):
        cid, n = struct.unpack_from(">ii", raw, pos)
        if cid != VEC_FILE_CLASSID:
            raise ValueError("not a PETSc Vec record at byte %d" % pos)
        pos += 8
        out.append(np.frombuffer(raw, dtype=">f8", count=n, offset=pos).astype(np.float64))
        pos += 8 * n
    return out


def write_restart(prefix, itr, fscale, x, xPhys, xo1, xo2, U, L):
    """TopOpt::WriteRestartFiles (TopOpt.cc:514-570): <prefix>.dat + <prefix>_itr_f0.dat"""
    write_petsc_vecs(prefix + ".dat", [x, xPhys, xo1, xo2, U, L])
    with open(prefix + "_itr_f0.dat", "w") as f:
        f.write("%d  %e\n" % (itr, fscale))


def read_restart(prefix):
    x, xPhys, xo1, xo2, U, L = read_petsc_vecs(prefix + ".dat")
    itr, fscale = open(prefix + "_itr_f0.dat").read().split()
    return int(itr), float(fscale), x, xPhys, xo1, xo2, U, L
