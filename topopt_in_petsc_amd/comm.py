"""z-slab neighbour exchange and scalar all-reduce through torch.distributed.

One process per GPU.  With the `nccl` backend (= RCCL over xGMI on ROCm) the
staging tensors are sent device-to-device; with `gloo` (CPU tests, or two ranks
sharing one GPU in the 1-GPU parity test) they are staged through host memory.
The HIP library calls back into `exchange` / `allreduce_sum`
(include/topopt_amd.h: tp_comm); both are ordered on the current stream.

The same class drives CPU tensors, which is what the world_size-2 gloo tests
use to check the partition/halo logic without a GPU.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import lib as _lib


class SlabComm:
    def __init__(self, part, device, group=None, cap=None):
        self.part = part
        self.device = torch.device(device)
        self.group = group
        self.rank, self.nranks = part.rank, part.nranks
        assert dist.is_initialized(), "torch.distributed must be initialised for nranks > 1"
        self.backend = dist.get_backend(group)
        # The library orders its kernels on the stream it was created with (api.Grid sets this attribute); the
        # C callbacks below issue their collectives under that SAME stream, whatever stream is current when the
        # library calls back (a Grid built inside a torch.cuda.stream(...) block and used outside it, or the reverse,
        # would otherwise race on the staging buffers).
        self.stream = None
        # staging: large enough for one 3-dof node plane and a few filter layers
        self.cap = int(cap or max(3 * part.plane, 4 * part.ex * part.ey, 1 << 20))
        mk = lambda n: torch.zeros(n, dtype=torch.float64, device=self.device)
        self.send_lo, self.send_hi, self.recv_lo, self.recv_hi = mk(self.cap), mk(self.cap), mk(self.cap), mk(self.cap)
        self.red = mk(16)
        self.gather = mk(self.cap * self.nranks)   # replicated coarsest level: all-gather target
        self.n_allgathers = 0
        self.n_exchanges = 0
        self._ops_cache = {}
        self.n_allreduces = 0
        self.bytes_sent = 0
        self._via_host = self.device.type == "cuda" and self.backend != "nccl"
        # the first collective of a group must involve all its ranks (the later neighbour exchanges do not)
        dist.all_reduce(self.red[:1] if not self._via_host else torch.zeros(1, dtype=torch.float64), group=self.group)
        # keep the callbacks alive for the lifetime of the object
        self._ex_cb = _lib.EXCHANGE_FN(self._exchange_cb)
        self._ar_cb = _lib.ALLREDUCE_FN(self._allreduce_cb)
        self._ag_cb = _lib.EXCHANGE_FN(self._allgather_cb)
        self._dx_cb = _lib.DIRECT_FN(self._direct_cb)
        self._ss_cb = _lib.SETSTREAM_FN(self._set_stream_cb)
        self.home_stream = None
        self._raw_cache = {}
        self._direct_cache = {}
        self.n_direct = 0
        p = lambda t: t.data_ptr()
        self.c_struct = _lib.Comm(None, p(self.send_lo), p(self.send_hi), p(self.recv_lo), p(self.recv_hi),
                                  p(self.red), self.cap, self._ex_cb, self._ar_cb, p(self.gather), self._ag_cb,
                                  self._dx_cb if self.device.type == "cuda" else _lib.DIRECT_FN(0),
                                  _lib.INPLACE_FN(0),   # in-place reductions: only the in-library RCCL path has them
                                  self._ss_cb if self.device.type == "cuda" else _lib.SETSTREAM_FN(0))

    # ---- python-level API (also used directly by the CPU tests) -----------
    def exchange(self, n):
        """send_lo[:n] -> rank-1, send_hi[:n] -> rank+1; recv_lo <- rank-1, recv_hi <- rank+1"""
        lo, hi = self.rank - 1, self.rank + 1
        has_lo, has_hi = lo >= 0, hi < self.nranks
        if self._via_host:
            s_lo, s_hi = self.send_lo[:n].cpu(), self.send_hi[:n].cpu()
            r_lo, r_hi = torch.empty(n, dtype=torch.float64), torch.empty(n, dtype=torch.float64)
        else:
            s_lo, s_hi, r_lo, r_hi = self.send_lo[:n], self.send_hi[:n], self.recv_lo[:n], self.recv_hi[:n]
        ops = None if self._via_host else self._ops_cache.get(n)
        if ops is None:
            ops = []
            if has_lo:
                ops += [dist.P2POp(dist.isend, s_lo, self._g(lo), self.group), dist.P2POp(dist.irecv, r_lo, self._g(lo), self.group)]
            if has_hi:
                ops += [dist.P2POp(dist.isend, s_hi, self._g(hi), self.group), dist.P2POp(dist.irecv, r_hi, self._g(hi), self.group)]
            if not self._via_host:
                self._ops_cache[n] = ops   # the staging tensors never move: the op list can be reused
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if self._via_host:
            if has_lo:
                self.recv_lo[:n].copy_(r_lo)
            if has_hi:
                self.recv_hi[:n].copy_(r_hi)
        self.n_exchanges += 1
        self.bytes_sent += 8 * n * (int(has_lo) + int(has_hi))

    # ---- zero-copy halo: the library's own vectors wrapped as tensors (no staging copies) ----
    def _raw(self, ptr, n):
        """a float64 tensor view of n doubles of device memory owned by the HIP library"""
        key = (ptr, n)
        t = self._raw_cache.get(key)
        if t is None:
            class _Arr:
                pass
            a = _Arr()
            a.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2,
                                          "strides": None}
            t = torch.as_tensor(a, device=self.device)
            assert t.data_ptr() == ptr and t.numel() == n and t.dtype == torch.float64
            self._raw_cache[key] = t
        return t

    def _direct_prepare(self, to_lo, from_lo, to_hi, from_hi, n):
        lo, hi = self.rank - 1, self.rank + 1
        has_lo, has_hi = lo >= 0, hi < self.nranks
        ts = [self._raw(p, n) if (p and ok) else None
              for p, ok in ((to_lo, has_lo), (from_lo, has_lo), (to_hi, has_hi), (from_hi, has_hi))]
        ops = []
        if not self._via_host:
            if has_lo:
                ops += [dist.P2POp(dist.isend, ts[0], self._g(lo), self.group),
                        dist.P2POp(dist.irecv, ts[1], self._g(lo), self.group)]
            if has_hi:
                ops += [dist.P2POp(dist.isend, ts[2], self._g(hi), self.group),
                        dist.P2POp(dist.irecv, ts[3], self._g(hi), self.group)]
        ent = (ts, ops)
        self._direct_cache[(to_lo, from_lo, to_hi, from_hi, n)] = ent
        return ent

    def exchange_direct(self, to_lo, from_lo, to_hi, from_hi, n):
        lo, hi = self.rank - 1, self.rank + 1
        has_lo, has_hi = lo >= 0, hi < self.nranks
        ent = self._direct_cache.get((to_lo, from_lo, to_hi, from_hi, n)) or \
            self._direct_prepare(to_lo, from_lo, to_hi, from_hi, n)
        ts, ops = ent
        if self._via_host:  # gloo with GPU memory: same tensors, staged through the host
            hops, r_lo, r_hi = [], None, None
            if has_lo:
                r_lo = torch.empty(n, dtype=torch.float64)
                hops += [dist.P2POp(dist.isend, ts[0].cpu(), self._g(lo), self.group),
                         dist.P2POp(dist.irecv, r_lo, self._g(lo), self.group)]
            if has_hi:
                r_hi = torch.empty(n, dtype=torch.float64)
                hops += [dist.P2POp(dist.isend, ts[2].cpu(), self._g(hi), self.group),
                         dist.P2POp(dist.irecv, r_hi, self._g(hi), self.group)]
            for w in dist.batch_isend_irecv(hops):
                w.wait()
            if has_lo:
                ts[1].copy_(r_lo)
            if has_hi:
                ts[3].copy_(r_hi)
        elif ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        self.n_direct += 1
        self.bytes_sent += 8 * n * (int(has_lo) + int(has_hi))

    def _direct_cb(self, _user, to_lo, from_lo, to_hi, from_hi, n):
        args = (to_lo or 0, from_lo or 0, to_hi or 0, from_hi or 0, int(n))
        if args not in self._direct_cache:
            try:  # wrapping the library's pointers happens before any communication: a failure here is recoverable
                self._direct_prepare(*args)
            except Exception as e:
                print("SlabComm: zero-copy halo unavailable (%r); staging buffers are used instead" % (e,), flush=True)
                return 2   # the library falls back to exchange() for good (same message sizes and order)
        try:
            with self._on_stream():
                self.exchange_direct(*args)
            return 0
        except Exception as e:
            print("SlabComm.exchange_direct failed: %r" % (e,), flush=True)
            return 1

    def allreduce_sum(self, n):
        if self._via_host:
            t = self.red[:n].cpu()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self.red[:n].copy_(t)
        else:
            dist.all_reduce(self.red[:n], op=dist.ReduceOp.SUM, group=self.group)
        self.n_allreduces += 1

    def allgather(self, n):
        """gather[r*n:(r+1)*n] <- rank r's send_lo[:n]"""
        out = self.gather[: n * self.nranks]
        if self._via_host:
            src = self.send_lo[:n].cpu()
            parts = [torch.empty(n, dtype=torch.float64) for _ in range(self.nranks)]
            dist.all_gather(parts, src, group=self.group)
            out.copy_(torch.cat(parts))
        elif self.backend == "nccl":
            dist.all_gather_into_tensor(out, self.send_lo[:n], group=self.group)
        else:
            parts = list(out.view(self.nranks, n).unbind(0))
            dist.all_gather(parts, self.send_lo[:n].contiguous(), group=self.group)
        self.n_allgathers += 1

    def _g(self, r):
        """group rank -> global rank"""
        return r if self.group is None else dist.get_global_rank(self.group, r)

    # ---- C callbacks --------------------------------------------------------
    def _on_stream(self):
        import contextlib
        if self.stream is not None and self.device.type == "cuda":
            return torch.cuda.stream(self.stream)
        return contextlib.nullcontext()

    def _set_stream_cb(self, _user, stream):
        """set_stream hook: the library's second (halo) stream, or NULL = back to the grid's stream"""
        try:
            if self.home_stream is None:
                self.home_stream = self.stream
            self.stream = torch.cuda.ExternalStream(stream, device=self.device) if stream else self.home_stream
        except Exception as e:
            print("SlabComm.set_stream failed: %r" % (e,), flush=True)

    def _exchange_cb(self, _user, n):
        try:
            with self._on_stream():
                self.exchange(int(n))
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("SlabComm.exchange failed: %r" % (e,), flush=True)
            return 1

    def _allgather_cb(self, _user, n):
        try:
            with self._on_stream():
                self.allgather(int(n))
            return 0
        except Exception as e:
            print("SlabComm.allgather failed: %r" % (e,), flush=True)
            return 1

    def _allreduce_cb(self, _user, n):
        try:
            with self._on_stream():
                self.allreduce_sum(int(n))
            return 0
        except Exception as e:
            print("SlabComm.allreduce_sum failed: %r" % (e,), flush=True)
            return 1

    # ---- host-side helpers on plain tensors (CPU or GPU) ----------------------
    def halo_nodes(self, v, dof):
        """refresh the ghost planes of a local node vector in place"""
        p = self.part
        pl = p.plane * dof
        if self.nranks == 1:
            return v
        if p.has_lo:
            self.send_lo[:pl].copy_(v[pl * p.own_lo: pl * (p.own_lo + 1)])
        if p.has_hi:
            self.send_hi[:pl].copy_(v[pl * p.own_hi: pl * (p.own_hi + 1)])
        self.exchange(pl)
        if p.has_lo:
            v[:pl].copy_(self.recv_lo[:pl])
        if p.has_hi:
            v[pl * (p.nz_local - 1): pl * p.nz_local].copy_(self.recv_hi[:pl])
        return v

    def dot_owned(self, a, b, dof):
        s = self.part.owned_slice(dof)
        self.red[0] = torch.dot(a[s], b[s])
        if self.nranks > 1:
            self.allreduce_sum(1)
        return float(self.red[0])
