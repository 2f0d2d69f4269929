"""z-slab partition of the structured hex grid (host logic, no GPU needed).

Mirrors the DMDA ownership rule the reference relies on
(LinearElasticity.cc:802-814: a rank owns the elements whose upper corner node
it owns): rank r owns element layers [r*ezl, (r+1)*ezl) and node planes
(r*ezl, (r+1)*ezl]; rank 0 additionally owns plane 0.  The local node array
stores planes r*ezl .. (r+1)*ezl (+1 ghost above unless last rank).
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class SlabPartition:
    nx: int  # global node counts
    ny: int
    nz: int
    rank: int = 0
    nranks: int = 1

    def __post_init__(self):
        if (self.nz - 1) % self.nranks:
            raise ValueError("nz-1 = %d element layers not divisible by %d ranks" % (self.nz - 1, self.nranks))

    # -- elements ---------------------------------------------------------
    @property
    def ex(self):
        return self.nx - 1

    @property
    def ey(self):
        return self.ny - 1

    @property
    def ez(self):
        return self.nz - 1

    @property
    def ez_own(self):
        return self.ez // self.nranks

    @property
    def elem_z0(self):
        return self.rank * self.ez_own

    @property
    def n_own_elems(self):
        return self.ex * self.ey * self.ez_own

    # -- nodes ------------------------------------------------------------
    @property
    def has_lo(self):
        return self.rank > 0

    @property
    def has_hi(self):
        return self.rank < self.nranks - 1

    @property
    def node_z0(self):
        """global index of local node plane 0"""
        return self.rank * self.ez_own

    @property
    def nz_local(self):
        return self.ez_own + 1 + (1 if self.has_hi else 0)

    @property
    def own_lo(self):
        return 1 if self.has_lo else 0

    @property
    def own_hi(self):
        return self.ez_own

    @property
    def plane(self):
        return self.nx * self.ny

    @property
    def n_local_nodes(self):
        return self.plane * self.nz_local

    @property
    def n_owned_nodes(self):
        return self.plane * (self.own_hi - self.own_lo + 1)

    def owned_slice(self, dof=1):
        """slice of the local node array (times dof) holding the owned planes"""
        return slice(self.plane * self.own_lo * dof, self.plane * (self.own_hi + 1) * dof)

    def global_slice(self, dof=1):
        """slice of the GLOBAL node array covered by this rank's local array (incl. ghosts)"""
        return slice(self.plane * self.node_z0 * dof, self.plane * (self.node_z0 + self.nz_local) * dof)

    def global_elem_slice(self):
        lay = self.ex * self.ey
        return slice(lay * self.elem_z0, lay * (self.elem_z0 + self.ez_own))

    def coarsenable(self, nlvls):
        f = 1 << (nlvls - 1)
        return self.ex % f == 0 and self.ey % f == 0 and self.ez_own % f == 0

    def level(self, l):
        """partition of multigrid level l (factor-2 coarsening, TopOpt.cc:183-201)"""
        f = 1 << l
        if self.ex % f or self.ey % f or self.ez_own % f:
            raise ValueError("level %d does not exist for this mesh" % l)
        return SlabPartition(self.ex // f + 1, self.ey // f + 1, self.ez // f + 1, self.rank, self.nranks)
