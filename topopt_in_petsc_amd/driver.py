"""The reference's optimisation loop (main.cc:22-141) on top of the MI355X hot path.

`TopOpt` carries the reference's parameters and defaults (TopOpt.cc:102-144) and
`run()` repeats main.cc's STEP 7 loop body:  solve + sensitivities -> objective
scaling -> Filter::Gradients -> move limits -> MMA -> design change -> (beta
continuation) -> FilterProject -> MND, printing the reference's per-iteration line.
Everything stays on the device; the host sees scalars only.
"""
import time
from dataclasses import dataclass, field

import torch

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

    def _increase_beta(self, gx, ch):
        """Filter::IncreaseBeta, Filter.cc:268-288"""
        if (ch < 0.01 or self.itr % 10 == 0) and self.beta < self.betaFinal and gx < 0.000001:
            self.beta = self.beta + 1 if self.beta < 7 else self.beta * 1.2
            self.beta = min(self.beta, self.betaFinal)

    def run(self, max_itr=None, verbose=False):
        ch = 1.0
        n = self.maxItr if max_itr is None else max_itr
        while self.itr < n and ch > 0.01:                                                        # main.cc:54
            ch = self.step(verbose)["ch"]
        return self.history
