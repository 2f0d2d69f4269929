"""The reference's optimisation loop (main.cc:22-141) on top of the MI355X hot path.

`TopOpt` carries the reference's parameters and defaults (TopOpt.cc:102-144) and
`run()` repeats main.cc's STEP 7 loop body:  solve + sensitivities -> objective
scaling -> Filter::Gradients -> move limits -> MMA -> design change -> (beta
continuation) -> FilterProject -> MND, printing the reference's per-iteration line.
Everything stays on the device; the host sees scalars only.
"""
import os
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import mpiio
from .api import Filter, Grid, LinearElasticity, MMA, SolverOptions


@dataclass
class TopOpt:
    # mesh (TopOpt.cc:106-116; nxyz are NODE counts like -nx -ny -nz)
    nxyz: tuple = (65, 33, 33)
    xc: tuple = (0.0, 2.0, 0.0, 1.0, 0.0, 1.0)
    nu: float = 0.3
    nlvls: int = 4
    # optimisation (TopOpt.cc:118-135)
    volfrac: float = 0.12
    maxItr: int = 400
    rmin: float = 0.08
    penal: float = 3.0
    Emin: float = 1.0e-9
    Emax: float = 1.0
    filter: int = 1
    Xmin: float = 0.0
    Xmax: float = 1.0
    movlim: float = 0.2
    projectionFilter: bool = False
    beta: float = 0.1
    betaFinal: float = 48.0
    eta: float = 0.0
    m: int = 1
    rank: int = 0
    nranks: int = 1
    solver: SolverOptions = None
    # I/O around the loop (main.cc:40, :113-129; TopOpt.cc:400-512): off unless a workdir is given
    workdir: str = None
    output: bool = True            # output_00000.dat (MPIIO container)
    restart: bool = True           # Restart0x.dat / Restart0x_itr_f0.dat / RestartSol0x.dat, alternating
    restartFileVec: str = None     # -restartFileVec / -restartFileItr: continue from these
    restartFileItr: str = None
    restartFileVecSol: str = None  # -restartFileVecSol: the state U (LinearElasticity.cc:590-606)
    onlyLoadDesign: bool = False
    history: list = field(default_factory=list)

    def __post_init__(self):
        nx, ny, nz = self.nxyz
        h = ((self.xc[1] - self.xc[0]) / (nx - 1), (self.xc[3] - self.xc[2]) / (ny - 1),
             (self.xc[5] - self.xc[4]) / (nz - 1))
        self.grid = Grid(nx, ny, nz, h, rank=self.rank, nranks=self.nranks)
        so = self.solver or SolverOptions(nlvls=self.nlvls, nu=self.nu)
        self.physics = LinearElasticity(self.grid, so)
        self.physics.SetUpLoadAndBC()
        self.filt = Filter(self.grid, self.filter, self.rmin)
        g = self.grid
        # TopOpt.cc:362-381: all design fields start at volfrac
        self.x = g.elem_vec(self.volfrac)
        self.xTilde, self.xPhys = g.elem_vec(self.volfrac), g.elem_vec(self.volfrac)
        self.dfdx, self.dgdx = g.elem_vec(), [g.elem_vec() for _ in range(self.m)]
        self.xmin, self.xmax, self.xold = g.elem_vec(), g.elem_vec(), g.elem_vec(self.volfrac)
        self.mma = MMA(g, self.x, self.m)
        self.fscale = 1.0
        self.itr = 0
        self._flip = True
        self._out = None
        if self.workdir is not None:
            os.makedirs(self.workdir, exist_ok=True)
            if self.output:
                self._out = mpiio.MPIIO(g.part, h, filename=os.path.join(self.workdir, "output_00000.dat"),
                                        xc0=(self.xc[0], self.xc[2], self.xc[4]))
        if self.restartFileVec and self.restartFileItr and os.path.exists(self.restartFileVec) \
                and os.path.exists(self.restartFileItr):
            self.ReadRestartFiles(self.restartFileVec, self.restartFileItr, self.restartFileVecSol)
        # main.cc:48
        self.filt.FilterProject(self.x, self.xTilde, self.xPhys, self.projectionFilter, self.beta, self.eta)

    def step(self, verbose=False):
        """one pass of the loop body, main.cc:54-111; returns the record of this iteration"""
        self.itr += 1
        t1 = time.perf_counter()
        fx, gx = self.physics.ComputeObjectiveConstraintsSensitivities(
            self.dfdx, self.dgdx[0], self.xPhys, self.Emin, self.Emax, self.penal, self.volfrac)   # main.cc:62
        if self.itr == 1:
            self.fscale = 10.0 / fx                                                              # :68-70
        fxs = fx * self.fscale
        self.dfdx.mul_(self.fscale)                                                              # :73
        self.filt.Gradients(self.x, self.xTilde, self.dfdx, self.dgdx, self.projectionFilter, self.beta, self.eta)
        self.mma.SetOuterMovelimit(self.Xmin, self.Xmax, self.movlim, self.x, self.xmin, self.xmax)  # :81
        self.mma.Update(self.x, self.dfdx, [gx], self.dgdx, self.xmin, self.xmax)                # :85
        ch = self.mma.DesignChange(self.x, self.xold)                                            # :89
        if self.projectionFilter:                                                                # :93-95
            self._increase_beta(gx, ch)
        self.filt.FilterProject(self.x, self.xTilde, self.xPhys, self.projectionFilter, self.beta, self.eta)  # :98
        mnd = self.filt.GetMND(self.xPhys)                                                       # :102
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rec = dict(itr=self.itr, fx=fx, fx_scaled=fxs, gx=gx, ch=ch, mnd=mnd, time=t2 - t1,
                   ksp_its=self.physics.last_its, ksp_rerr=self.physics.last_rnorm / self.physics.last_bnorm,
                   mma_inner=self.mma.last_inner)
        self.history.append(rec)
        if verbose and self.rank == 0:
            print("It.: %i, True fx: %f, Scaled fx: %f, gx[0]: %f, ch.: %f, mnd.: %f, time: %f"
                  % (self.itr, fx, fxs, gx, ch, mnd, t2 - t1), flush=True)                        # :108-111
        return rec

    # ---- restart / output (host-side I/O; vectors gathered in natural = rank order) ----
    def _gather(self, t):
        a = t.detach().cpu().numpy()
        if self.nranks == 1:
            return a
        import torch.distributed as dist
        parts = [None] * self.nranks
        dist.all_gather_object(parts, a)
        return np.concatenate(parts)

    def _own(self, a):
        n = self.grid.part.n_own_elems
        return torch.from_numpy(np.ascontiguousarray(a[self.rank * n:(self.rank + 1) * n])).to(self.x.device)

    def WriteRestartFiles(self):
        """TopOpt::WriteRestartFiles (TopOpt.cc:514-570) + LinearElasticity::WriteRestartFiles (:447-478)"""
        if self.workdir is None or not self.restart:
            return None
        self._flip = not self._flip
        tag = "01" if self._flip else "00"
        g = self.grid
        xo1, xo2, U, L = g.elem_vec(), g.elem_vec(), g.elem_vec(), g.elem_vec()
        self.mma.Restart(xo1, xo2, U, L)
        vecs = [self._gather(v) for v in (self.x, self.xPhys, xo1, xo2, U, L)]
        sol = self._gather(self.physics.U[self.grid.part.owned_slice(3)])
        prefix = os.path.join(self.workdir, "Restart" + tag)
        if self.rank == 0:
            mpiio.write_restart(prefix, self.itr, self.fscale, *vecs)
            mpiio.write_petsc_vecs(os.path.join(self.workdir, "RestartSol%s.dat" % tag), [sol])
        return prefix

    def ReadRestartFiles(self, vecfile, itrfile, solfile=None):
        """TopOpt::AllocateMMAwithRestart (TopOpt.cc:474-507); solfile = -restartFileVecSol (LinearElasticity.cc:590-606)"""
        x, xPhys, xo1, xo2, U, L = mpiio.read_petsc_vecs(vecfile)
        itr, fscale = open(itrfile).read().split()
        self.x.copy_(self._own(x))
        self.xPhys.copy_(self._own(xPhys))
        self.fscale = float(fscale)
        self.itr = int(itr)
        if not self.onlyLoadDesign:
            self.mma.SetRestart(self.itr, self._own(xo1), self._own(xo2), self._own(U), self._own(L))
        if solfile and os.path.exists(solfile):
            sol, = mpiio.read_petsc_vecs(solfile)
            self.physics.U.zero_()
            sl = self.grid.part.owned_slice(3)
            n0 = 3 * self.grid.part.plane * (self.grid.part.node_z0 + self.grid.part.own_lo)
            self.physics.U[sl] = torch.from_numpy(sol[n0:n0 + (sl.stop - sl.start)].copy()).to(self.x.device)

    def WriteVTK(self, itr):
        if self._out is not None:
            self._out.WriteVTK(self.physics.U, self.x, self.xTilde, self.xPhys, itr)

    def _increase_beta(self, gx, ch):
        """Filter::IncreaseBeta, Filter.cc:268-288"""
        if (ch < 0.01 or self.itr % 10 == 0) and self.beta < self.betaFinal and gx < 0.000001:
            self.beta = self.beta + 1 if self.beta < 7 else self.beta * 1.2
            self.beta = min(self.beta, self.betaFinal)

    def run(self, max_itr=None, verbose=False):
        ch = 1.0
        n = self.maxItr if max_itr is None else max_itr
        while self.itr < n and ch > 0.01:                                                        # main.cc:54
            beta0 = self.beta
            ch = self.step(verbose)["ch"]
            if self.itr < 11 or self.itr % 20 == 0 or self.beta != beta0:                        # :114-116
                self.WriteVTK(self.itr)
            if self.itr % 10 == 0:                                                               # :119-122
                self.WriteRestartFiles()
        self.WriteRestartFiles()                                                                 # :125-126
        self.WriteVTK(self.itr + 1)                                                              # :129
        return self.history
