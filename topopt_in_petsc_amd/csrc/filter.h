// filter.h -- density/sensitivity filter (Filter.cc) and Helmholtz PDE filter
// (PDEFilter.cc).  Included at the end of topopt_amd.hip.
//
// The reference builds an explicit sparse matrix H (nnz/row grows with
// (rmin/h)^3, Filter.cc:404-448).  The cone weights R - |c_i - c_j| are
// translation invariant on the structured grid, so here H is never stored: the
// filter is a (2c+1)^3 stencil with a small weight table, truncated at the
// domain boundary exactly like the reference's k2/j2/i2 loops (:417-420).
#pragma once

// out_i = (sum_j w_ij in_j) [/ d1_i] [/ d2_i];  in = ghosted copy (conn layers below/above the own ones)
__global__ __launch_bounds__(BLK) void k_conv_filter(int ex, int ey, int ez_own, int conn, int e0z, int ez_glob,
                                                     const double *__restrict__ xg, const double *__restrict__ wtab,
                                                     double *__restrict__ out, const double *__restrict__ d1,
                                                     const double *__restrict__ d2) {
    const long nel = (long)ex * ey * ez_own;
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= nel) return;
    const int i = (int)(t % ex), j = (int)((t / ex) % ey), k = (int)(t / ((long)ex * ey));
    const int w1 = 2 * conn + 1;
    const int klo = max(k + e0z - conn, 0) - e0z, khi = min(k + e0z + conn, ez_glob - 1) - e0z;
    const int jlo = max(j - conn, 0), jhi = min(j + conn, ey - 1);
    const int ilo = max(i - conn, 0), ihi = min(i + conn, ex - 1);
    double s = 0.0;
    for (int k2 = klo; k2 <= khi; k2++)
        for (int j2 = jlo; j2 <= jhi; j2++) {
            const double *__restrict__ row = xg + (long)ex * (j2 + (long)ey * (k2 + conn));
            const double *__restrict__ wr = wtab + ((k2 - k + conn) * w1 + (j2 - j + conn)) * w1 + (conn - i);
            for (int i2 = ilo; i2 <= ihi; i2++) s = fma(wr[i2], row[i2], s);
        }
    if (d1) s = s / d1[t];
    if (d2) s = s / d2[t];
    out[t] = s;
}
// The same filter as an LDS-tiled stencil: a 32 x 4 x 2 block of elements stages its (32+2C) x (4+2C) x (2+2C)
// neighbourhood once (zeros outside the domain), then every thread sums its (2C+1)^3 window out of LDS in the same
// order as k_conv_filter -- the skipped out-of-domain terms become exact zeros, so the results are bitwise the same.
// The weights are workgroup-uniform (scalar loads).  6.75 HBM/L2 loads per element instead of 125 (C = 2).
template <int C>
__global__ __launch_bounds__(256) void k_conv_filter_tiled(int ex, int ey, int ez_own, int e0z, int ez_glob,
                                                           const double *__restrict__ xg, const double *__restrict__ wtab,
                                                           double *__restrict__ out, const double *__restrict__ d1,
                                                           const double *__restrict__ d2) {
    constexpr int TX = 32, TY = 4, TZ = 2, W1 = 2 * C + 1, SX = TX + 2 * C, SY = TY + 2 * C, SZ = TZ + 2 * C;
    __shared__ double s_x[SZ * SY * SX];
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, z0 = blockIdx.z * TZ;
    for (int f = threadIdx.x; f < SZ * SY * SX; f += 256) {
        const int sx = f % SX, sy = (f / SX) % SY, sz = f / (SX * SY);
        const int gi = x0 - C + sx, gj = y0 - C + sy, kl = z0 - C + sz;  // kl: layer relative to the own range
        const bool ok = gi >= 0 && gi < ex && gj >= 0 && gj < ey && kl + e0z >= 0 && kl + e0z < ez_glob && kl < ez_own + C;
        s_x[f] = ok ? xg[(long)gi + (long)ex * (gj + (long)ey * (kl + C))] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x % TX, ty = (threadIdx.x / TX) % TY, tz = threadIdx.x / (TX * TY);
    const int i = x0 + tx, j = y0 + ty, k = z0 + tz;
    if (i >= ex || j >= ey || k >= ez_own) return;
    double s = 0.0;
#pragma unroll
    for (int dk = 0; dk < W1; dk++)
#pragma unroll
        for (int dj = 0; dj < W1; dj++)
#pragma unroll
            for (int di = 0; di < W1; di++)
                s = fma(wtab[(dk * W1 + dj) * W1 + di], s_x[((tz + dk) * SY + (ty + dj)) * SX + tx + di], s);
    const long t = (long)i + (long)ex * (j + (long)ey * k);
    if (d1) s = s / d1[t];
    if (d2) s = s / d2[t];
    out[t] = s;
}
// Round 6, small radii (ElemConn 1, 2; counters: the one-output form above is LDS-issue bound -- one ds_read per fma): NO outputs
// per thread along z.  A staged value serves up to NO outputs (those whose window holds its plane), so a thread reads
// (NO + 2C)(2C+1)^2 values for NO outputs instead of NO (2C+1)^3 -- 2.5 x fewer at C = 2, NO = 4 -- at unit lane stride in x (the
// four-outputs-along-x form conflicts in the LDS banks at these radii, below).  Every output is still ONE fma chain over
// (dk, dj, di) ascending with exact zeros outside the domain: the bits of k_conv_filter.  The weights are sign-symmetric bit for
// bit ((di dx)^2 ...), so the (C+1)^3 values of one octant are read once into scalar registers.
template <int C, int NO>
__global__ __launch_bounds__(256) void k_conv_filter_zmulti(int ex, int ey, int ez_own, int e0z, int ez_glob,
                                                            const double *__restrict__ xg, const double *__restrict__ wtab,
                                                            double *__restrict__ out, const double *__restrict__ d1,
                                                            const double *__restrict__ d2) {
    constexpr int TX = 32, TY = 4, TZT = 2, TZ = TZT * NO, W1 = 2 * C + 1, SX = TX + 2 * C, SY = TY + 2 * C, SZ = TZ + 2 * C;
    __shared__ double s_x[SZ * SY * SX];
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, z0 = blockIdx.z * TZ;
    for (int f = threadIdx.x; f < SZ * SY * SX; f += 256) {
        const int sx = f % SX, sy = (f / SX) % SY, sz = f / (SX * SY);
        const int gi = x0 - C + sx, gj = y0 - C + sy, kl = z0 - C + sz;  // kl: layer relative to the own range
        const bool ok = gi >= 0 && gi < ex && gj >= 0 && gj < ey && kl + e0z >= 0 && kl + e0z < ez_glob && kl < ez_own + C;
        s_x[f] = ok ? xg[(long)gi + (long)ex * (gj + (long)ey * (kl + C))] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x % TX, ty = (threadIdx.x / TX) % TY, tzt = threadIdx.x / (TX * TY);
    const int i = x0 + tx, j = y0 + ty, k0 = z0 + NO * tzt;
    if (i >= ex || j >= ey || k0 >= ez_own) return;
    double wq[C + 1][C + 1][C + 1];
#pragma unroll
    for (int a = 0; a <= C; a++)
#pragma unroll
        for (int b = 0; b <= C; b++)
#pragma unroll
            for (int c = 0; c <= C; c++) wq[a][b][c] = wtab[((C + a) * W1 + (C + b)) * W1 + (C + c)];
    double acc[NO];
#pragma unroll
    for (int o = 0; o < NO; o++) acc[o] = 0.0;
    // Straight-line code, scheduled row by row: left alone the compiler sinks the sums of the later outputs behind the last row
    // with every staged value live (370 VGPRs at C = 2, NO = 4: one wave per SIMD, 78 us against the one-output form's 49).  The
    // pins keep a row's fma of every output in the row's slot; the NEXT row is requested before the current one is consumed.
    constexpr int R = (NO + 2 * C) * W1;
    const double *__restrict__ base = s_x + ((NO * tzt) * SY + ty) * SX + tx;
    double v[W1], vn[W1];
#pragma unroll
    for (int di = 0; di < W1; di++) v[di] = base[di];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int p = r / W1, dj = r % W1;
        if (r + 1 < R) {
#pragma unroll
            for (int di = 0; di < W1; di++) vn[di] = base[(((r + 1) / W1) * SY + ((r + 1) % W1)) * SX + di];
        }
#pragma unroll
        for (int di = 0; di < W1; di++)
#pragma unroll
            for (int o = 0; o < NO; o++) {
                const int dk = p - o;
                if (dk >= 0 && dk < W1)
                    acc[o] = fma(wq[dk < C ? C - dk : dk - C][dj < C ? C - dj : dj - C][di < C ? C - di : di - C], v[di], acc[o]);
            }
#pragma unroll
        for (int o = 0; o < NO; o++) asm volatile("" : "+v"(acc[o]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int di = 0; di < W1; di++) v[di] = vn[di];
    }
#pragma unroll
    for (int o = 0; o < NO; o++) {
        const int k = k0 + o;
        if (k >= ez_own) break;
        const long t = (long)i + (long)ex * (j + (long)ey * k);
        double sres = acc[o];
        if (d1) sres = sres / d1[t];
        if (d2) sres = sres / d2[t];
        out[t] = sres;
    }
}
// Large radii (ElemConn 4 .. 8: the reference's own default rmin = 0.08 on the BASELINE meshes, TopOpt.cc:121 --
// 5 on 128x64x64 (1331 taps), 8 at 128^3 (4913)): the same tile idea with FOUR outputs per thread along x.  A thread
// loads a row of 4 + 2C staged values once and uses each of them for up to four outputs (sliding window): 3.2x fewer
// LDS reads per tap than k_conv_filter_tiled, whose 729+ reads per output would be the bound here.  Tile: 32 x TYE x
// TZE elements (8 x TYE x TZE threads), staged neighbourhood (32+2C)(TYE+2C)(TZE+2C) doubles: up to 139 KB of LDS.
// Every output is still one fma chain over (dk, dj, di) ascending with exact zeros outside the domain: the bits of
// k_conv_filter.  Only the innermost loop is unrolled (a full unroll of 4913 x 4 fma would not fit the I-cache);
// the weights are workgroup-uniform scalar loads.
template <int C, int TYE, int TZE>
__global__ __launch_bounds__(8 * TYE * TZE) void k_conv_filter_wide(int ex, int ey, int ez_own, int e0z, int ez_glob,
                                                                    const double *__restrict__ xg, const double *__restrict__ wtab,
                                                                    double *__restrict__ out, const double *__restrict__ d1,
                                                                    const double *__restrict__ d2) {
    constexpr int TXE = 32, W1 = 2 * C + 1, SX = TXE + 2 * C, SY = TYE + 2 * C, SZ = TZE + 2 * C, NT = 8 * TYE * TZE;
    __shared__ double s_x[SZ * SY * SX];
    const int x0 = blockIdx.x * TXE, y0 = blockIdx.y * TYE, z0 = blockIdx.z * TZE;
    for (int f = threadIdx.x; f < SZ * SY * SX; f += NT) {
        const int sx = f % SX, sy = (f / SX) % SY, sz = f / (SX * SY);
        const int gi = x0 - C + sx, gj = y0 - C + sy, kl = z0 - C + sz;
        const bool ok = gi >= 0 && gi < ex && gj >= 0 && gj < ey && kl + e0z >= 0 && kl + e0z < ez_glob && kl < ez_own + C;
        s_x[f] = ok ? xg[(long)gi + (long)ex * (gj + (long)ey * (kl + C))] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x % 8, ty = (threadIdx.x / 8) % TYE, tz = threadIdx.x / (8 * TYE);
    const int i0 = x0 + 4 * tx, j = y0 + ty, k = z0 + tz;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int dk = 0; dk < W1; dk++)
        for (int dj = 0; dj < W1; dj++) {
            const double *__restrict__ row = s_x + ((tz + dk) * SY + (ty + dj)) * SX + 4 * tx;
            const double *__restrict__ w = wtab + (dk * W1 + dj) * W1;
            double v[4 + 2 * C];
#pragma unroll
            for (int q = 0; q < 4 + 2 * C; q++) v[q] = row[q];
#pragma unroll
            for (int di = 0; di < W1; di++) {
                const double wv = w[di];
#pragma unroll
                for (int o = 0; o < 4; o++) acc[o] = fma(wv, v[o + di], acc[o]);
            }
        }
    if (j >= ey || k >= ez_own) return;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int i = i0 + o;
        if (i >= ex) break;
        const long t = (long)i + (long)ex * (j + (long)ey * k);
        double s = acc[o];
        if (d1) s = s / d1[t];
        if (d2) s = s / d2[t];
        out[t] = s;
    }
}
// ghosted input: mode 0: a, 1: a / b, 2: a * b
// Radii beyond ElemConn 8, up to 24 (round 4; the reference's absolute default rmin = 0.08 gives 10 at 128^3 and on C3, 20 on C5): the
// (32 + 2C) x (T + 2C) x (T + 2C) neighbourhood of a block no longer fits the LDS, so the z direction is STREAMED: a block of
// 32 x 16 x 4 elements (one wave per z layer, eight x-outputs per thread) stages one (32 + 2C) x (16 + 2C) plane at a time,
// double buffered, and every wave whose layer lies within C of the plane adds that plane's (2C+1)^2 taps to its sums -- the
// plane's row segment in registers, the weights of a (dk, dj) row wave-uniform (scalar loads).  An output sees its planes in
// ascending z and inside a plane the taps in the order of k_conv_filter, the skipped out-of-domain terms as exact zeros: the
// same bits as the direct form.  8 fma per LDS read: FP64 issue bounds it, (2C+1)^3 fma per element.
template <int C>
__global__ __launch_bounds__(256) void k_conv_filter_zring(int ex, int ey, int ez_own, int e0z, int ez_glob,
                                                           const double *__restrict__ xg, const double *__restrict__ wtab,
                                                           double *__restrict__ out, const double *__restrict__ d1,
                                                           const double *__restrict__ d2) {
    constexpr int TXE = 32, TYE = 16, TZE = 4, NX = 8, W1 = 2 * C + 1, SX = TXE + 2 * C, SY = TYE + 2 * C, NT = 256;
    // two planes of (16 + 2C) x (32 + 2C) doubles: 68.7 KB at C = 21, 81.9 KB at C = 24 -- sized for the 160 KB of a gfx950 CU
    // (this library is built for gfx950 only; a 64-KB-LDS part would have to stop at C = 20)
    static_assert(sizeof(double) * 2 * SY * SX <= 160 * 1024, "k_conv_filter_zring: the two staged planes exceed gfx950's 160 KB of LDS");
    __shared__ double s_x[2][SY * SX];
    const int x0 = blockIdx.x * TXE, y0 = blockIdx.y * TYE, z0 = blockIdx.z * TZE;
    const int tz = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);  // z layer of this WAVE
    const int tx = threadIdx.x & 3, ty = (threadIdx.x >> 2) & 15;
    auto stage = [&](int buf, int kl) {  // plane with local z index kl (zeros outside the domain / the ghosted copy)
        const bool zok = kl + e0z >= 0 && kl + e0z < ez_glob && kl < ez_own + C && kl >= -C;
        for (int f = threadIdx.x; f < SY * SX; f += NT) {
            const int sx = f % SX, sy = f / SX;
            const int gi = x0 - C + sx, gj = y0 - C + sy;
            const bool ok = zok && gi >= 0 && gi < ex && gj >= 0 && gj < ey;
            s_x[buf][f] = ok ? xg[(long)gi + (long)ex * (gj + (long)ey * (kl + C))] : 0.0;
        }
    };
    double acc[NX];
#pragma unroll
    for (int o = 0; o < NX; o++) acc[o] = 0.0;
    const int p_lo = z0 - C, p_hi = z0 + TZE - 1 + C;
    stage(0, p_lo);
    __syncthreads();
    for (int p = p_lo; p <= p_hi; p++) {
        const int buf = (p - p_lo) & 1;
        if (p < p_hi) stage(buf ^ 1, p + 1);   // (the other buffer: its last readers passed the barrier at the end of the previous trip)
        const int dk = p - (z0 + tz) + C;      // wave-uniform
        if (dk >= 0 && dk < W1) {
            for (int dj = 0; dj < W1; dj++) {
                const double *__restrict__ row = s_x[buf] + (ty + dj) * SX + NX * tx;
                const double *__restrict__ w = wtab + (dk * W1 + dj) * W1;
                double v[NX + 2 * C];
#pragma unroll
                for (int q = 0; q < NX + 2 * C; q++) v[q] = row[q];
#pragma unroll
                for (int di = 0; di < W1; di++) {
                    const double wv = w[di];
#pragma unroll
                    for (int o = 0; o < NX; o++) acc[o] = fma(wv, v[o + di], acc[o]);
                }
            }
        }
        __syncthreads();
    }
    const int j = y0 + ty, k = z0 + tz;
    if (j >= ey || k >= ez_own) return;
#pragma unroll
    for (int o = 0; o < NX; o++) {
        const int i = x0 + NX * tx + o;
        if (i >= ex) break;
        const long t = (long)i + (long)ex * (j + (long)ey * k);
        double s = acc[o];
        if (d1) s = s / d1[t];
        if (d2) s = s / d2[t];
        out[t] = s;
    }
}

__global__ __launch_bounds__(BLK) void k_fill_pw(double *__restrict__ y, const double *__restrict__ a,
                                                 const double *__restrict__ b, int mode, long n) {
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK)
        y[i] = mode == 0 ? a[i] : (mode == 1 ? a[i] / b[i] : a[i] * b[i]);
}
// Heaviside projection and its derivative (Filter.h:80-88)
__global__ __launch_bounds__(BLK) void k_heaviside(double *__restrict__ y, const double *__restrict__ x, double beta,
                                                   double eta, long n) {
    const double den = tanh(beta * eta) + tanh(beta * (1.0 - eta)), t0 = tanh(beta * eta);
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK)
        y[i] = (t0 + tanh(beta * (x[i] - eta))) / den;
}
// df *= d/dx~ projection (Filter.cc:125-164)
__global__ __launch_bounds__(BLK) void k_heaviside_chain(double *__restrict__ df, const double *__restrict__ xt,
                                                         double beta, double eta, long n) {
    const double den = tanh(beta * eta) + tanh(beta * (1.0 - eta));
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        const double th = tanh(beta * (xt[i] - eta));
        df[i] = df[i] * (beta * (1.0 - th * th) / den);
    }
}
__global__ __launch_bounds__(BLK) void k_mnd(const double *__restrict__ x, long n, double *__restrict__ partials) {
    double s = 0.0;
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) s += 4 * x[i] * (1.0 - x[i]);
    s = block_sum(s);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
// clamp to [0,1], count violations beyond 1e-4 (Filter.cc:76-100)
__global__ __launch_bounds__(BLK) void k_clamp01(double *__restrict__ x, long n, double *__restrict__ partials) {
    double viol = 0.0;
    for (long i = blockIdx.x * (long)BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        double v = x[i];
        if (v < 0.0) {
            if (fabs(v) > 1.0e-4) viol += 1.0;
            v = 0.0;
        }
        if (v > 1.0) {
            if (fabs(v - 1.0) > 1.0e-4) viol += 1.0;
            v = 1.0;
        }
        x[i] = v;
    }
    viol = block_sum(viol);
    if (threadIdx.x == 0) partials[blockIdx.x] = viol;
}

// PDE filter transfers, T = 1/8 element -> node (PDEFilter.cc:259, :567-575):
// rhs_n = vol * sum_e 0.125 x_e, u_n = sum_e 0.125 x_e (initial guess, :198-202)
__global__ __launch_bounds__(BLK) void k_pde_elem_to_node(Geom g, const double *__restrict__ xe, double vol,
                                                          double *__restrict__ rhs, double *__restrict__ u) {
    const long plane = g.plane();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= g.owned_nodes()) return;
    const int k = g.own_lo + (int)(t / plane);
    const int rem = (int)(t % plane);
    const int j = rem / g.nx, i = rem % g.nx;
    const long n = t + plane * g.own_lo;
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const int ei = i - LXc(a), ej = j - LYc(a), ek = k - LZc(a);
        if (ei < 0 || ei >= g.ex || ej < 0 || ej >= g.ey || ek < 0 || ek >= g.ezl) continue;
        s += 0.125 * xe[(long)ei + (long)g.ex * (ej + (long)g.ey * ek)];
    }
    u[n] = s;
    rhs[n] = s * vol;
}
// x~_e = sum_a 0.125 u_a   (MatMultTranspose(T), PDEFilter.cc:210)
__global__ __launch_bounds__(BLK) void k_pde_node_to_elem(Geom g, const double *__restrict__ u,
                                                          double *__restrict__ out) {
    const long nel = g.own_elems();
    const long t = blockIdx.x * (long)BLK + threadIdx.x;
    if (t >= nel) return;
    const int i = (int)(t % g.ex), j = (int)((t / g.ex) % g.ey), k = (int)(t / ((long)g.ex * g.ey));
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 8; a++) s += 0.125 * u[(long)(i + LXc(a)) + (long)g.nx * ((j + LYc(a)) + (long)g.ny * (k + LZc(a)))];
    out[t] = s;
}

struct tp_filter {
    tp_grid *grid;
    int type, conn;
    double R;
    long nel, lay;
    double *wtab, *Hs, *xg, *tmp;
    // PDE filter
    MGSolver<1> *pde;
    std::vector<double> KF;  // per level 64
    double *d_KF, *d_wtab = nullptr, *xe, *rhs, *u;
    double elemVol;
    int last_its;
    double last_rnorm;
    long violations;
};

static int filter_conv(tp_filter *f, double *out, const double *d1, const double *d2) {
    tp_grid *g = f->grid;
    const int c = f->conn;
    // ghost layers: own first c layers -> lower neighbour's top ghosts, own last c -> upper's bottom ghosts
    TP_TRY(exchange_segments(g, f->xg + c * f->lay, f->xg, f->xg + (long)g->ez_own * f->lay,
                             f->xg + (long)(c + g->ez_own) * f->lay, c * f->lay, 1, c * f->lay));
    const dim3 tg((g->ex + 31) / 32, (g->ey + 3) / 4, (g->ez_own + 1) / 2);
    static const bool no_tile = getenv("TP_NO_FILTER_TILE") != nullptr;
#define TP_CONV_TILED(CC)                                                                                            \
    TP_LAUNCH(k_conv_filter_tiled<CC>, tg, dim3(256), 0, g->stream, g->ex, g->ey, g->ez_own, g->rank * g->ez_own, \
                       g->ez_glob, f->xg, f->wtab, out, d1, d2)
#define TP_CONV_WIDE(CC, TYE, TZE)                                                                                                   \
    TP_LAUNCH((k_conv_filter_wide<CC, TYE, TZE>), dim3((g->ex + 31) / 32, (g->ey + TYE - 1) / TYE, (g->ez_own + TZE - 1) / TZE), \
              dim3(8 * TYE * TZE), 0, g->stream, g->ex, g->ey, g->ez_own, g->rank * g->ez_own, g->ez_glob, f->xg, f->wtab, out, d1, d2)
    // Round 6 (counters: the one-output form is LDS-issue bound, one ds_read per fma): the four-outputs-per-thread form was
    // measured for the small radii too, bit-equal -- ElemConn 3 (343 taps): 105.5 -> 69.8 us at 128^3, taken; ElemConn 2 (125
    // taps, the bench's 2.56 h): 96 us against 50 (its 32-byte lane stride of the staged rows conflicts in the LDS banks, and a
    // quarter of the threads), fully unrolled no better.  Several outputs per thread along Z instead keep the unit lane stride:
    // k_conv_filter_zmulti serves ElemConn 1 and 2 on all but the small meshes
    // outputs per thread along z at ElemConn 1, 2 (k_conv_filter_zmulti, bit-equal): by the number of workgroups the one-output
    // form would launch -- 128^3: 49.6 -> 36.1 us with four (two: 39.6), 128x64x64: 18.2 -> 15.6 with two (four: 16.4),
    // 48x24x24: 8.0 as it is (8.4 / 10.7); TP_FILTER_ZMULTI=0 / 2 / 4 forces one
    static const int zm_env = getenv("TP_FILTER_ZMULTI") ? atoi(getenv("TP_FILTER_ZMULTI")) : -1;
    const long wgs1 = (long)tg.x * tg.y * tg.z;
    const int zm = zm_env >= 0 ? zm_env : (wgs1 >= 8192 ? 4 : (wgs1 >= 1024 ? 2 : 0));
#define TP_CONV_ZMULTI(CC, NO)                                                                                                     \
    TP_LAUNCH((k_conv_filter_zmulti<CC, NO>), dim3((g->ex + 31) / 32, (g->ey + 3) / 4, (g->ez_own + 2 * NO - 1) / (2 * NO)), dim3(256), 0, \
              g->stream, g->ex, g->ey, g->ez_own, g->rank * g->ez_own, g->ez_glob, f->xg, f->wtab, out, d1, d2)
    if (!no_tile && c == 2 && zm == 4)
        TP_CONV_ZMULTI(2, 4);
    else if (!no_tile && c == 2 && zm == 2)
        TP_CONV_ZMULTI(2, 2);
    else if (!no_tile && c == 1 && zm >= 2)
        TP_CONV_ZMULTI(1, 4);  // (128^3: 25.9 -> 22.2 us)
    else if (!no_tile && c == 1)
        TP_CONV_TILED(1);
    else if (!no_tile && c == 2)
        TP_CONV_TILED(2);
    else if (!no_tile && c == 3)
        TP_CONV_WIDE(3, 8, 4);
    else if (!no_tile && c == 4)
        TP_CONV_WIDE(4, 8, 4);
    else if (!no_tile && c == 5)
        TP_CONV_WIDE(5, 8, 4);
    else if (!no_tile && c == 6)
        TP_CONV_WIDE(6, 8, 2);
    else if (!no_tile && c == 7)
        TP_CONV_WIDE(7, 8, 2);
    else if (!no_tile && c == 8)
        TP_CONV_WIDE(8, 4, 2);
#undef TP_CONV_WIDE
#undef TP_CONV_ZMULTI
#define TP_CONV_ZRING(CC)                                                                                                             \
    TP_LAUNCH((k_conv_filter_zring<CC>), dim3((g->ex + 31) / 32, (g->ey + 15) / 16, (g->ez_own + 3) / 4), dim3(256), 0, g->stream, \
              g->ex, g->ey, g->ez_own, g->rank * g->ez_own, g->ez_glob, f->xg, f->wtab, out, d1, d2)
    else if (!no_tile && c == 9)
        TP_CONV_ZRING(9);
    else if (!no_tile && c == 10)
        TP_CONV_ZRING(10);
    else if (!no_tile && c == 11)
        TP_CONV_ZRING(11);
    else if (!no_tile && c == 12)
        TP_CONV_ZRING(12);
    else if (!no_tile && c == 13)
        TP_CONV_ZRING(13);
    else if (!no_tile && c == 14)
        TP_CONV_ZRING(14);
    else if (!no_tile && c == 15)
        TP_CONV_ZRING(15);
    else if (!no_tile && c == 16)
        TP_CONV_ZRING(16);
    else if (!no_tile && c == 17)
        TP_CONV_ZRING(17);
    else if (!no_tile && c == 18)
        TP_CONV_ZRING(18);
    else if (!no_tile && c == 19)
        TP_CONV_ZRING(19);
    else if (!no_tile && c == 20)
        TP_CONV_ZRING(20);
    else if (!no_tile && c == 21)
        TP_CONV_ZRING(21);
    else if (!no_tile && c == 22)
        TP_CONV_ZRING(22);
    else if (!no_tile && c == 23)
        TP_CONV_ZRING(23);
    else if (!no_tile && c == 24)
        TP_CONV_ZRING(24);
#undef TP_CONV_ZRING
    else
        TP_LAUNCH(k_conv_filter, dim3((int)((f->nel + BLK - 1) / BLK)), dim3(BLK), 0, g->stream, g->ex, g->ey,
                           g->ez_own, c, g->rank * g->ez_own, g->ez_glob, f->xg, f->wtab, out, d1, d2);
#undef TP_CONV_TILED
    const double w3 = (2.0 * c + 1) * (2.0 * c + 1) * (2.0 * c + 1);
    count_launch(g, (16.0 + (d1 ? 8.0 : 0.0) + (d2 ? 8.0 : 0.0)) * f->nel, 2.0 * w3 * f->nel);
    return TP_OK;
}
static int filter_fill(tp_filter *f, const double *a, const double *b, int mode) {
    tp_grid *g = f->grid;
    TP_LAUNCH(k_fill_pw, dim3(grid_for(f->nel)), dim3(BLK), 0, g->stream, f->xg + f->conn * f->lay, a, b, mode,
                       f->nel);
    count_launch(g, (mode ? 24.0 : 16.0) * f->nel, mode ? 1.0 * f->nel : 0.0);
    return TP_OK;
}

// x~ = T^T K_f^-1 (vol T x)  (PDEFilt::FilterProject, PDEFilter.cc:189-216); in/out may alias
static int pde_apply(tp_filter *f, const double *in, double *out) {
    tp_grid *g = f->grid;
    MGSolver<1> &mg = *f->pde;
    Geom q = mg.lv[0].g;
    hipStream_t s = g->stream;
    TP_HIP(hipMemcpyAsync(f->xe, in, sizeof(double) * (size_t)f->nel, hipMemcpyDeviceToDevice, s));
    TP_TRY(exchange_segments(g, f->xe, nullptr, nullptr, f->xe + f->nel, f->lay, 1, f->lay));
    TP_LAUNCH(k_pde_elem_to_node, dim3((int)((q.owned_nodes() + BLK - 1) / BLK)), dim3(BLK), 0, s, q, f->xe,
                       f->elemVol, f->rhs, f->u);
    count_launch(g, 8.0 * f->nel + 16.0 * q.owned_nodes(), 9.0 * q.owned_nodes());
    int rc = mg.solve(f->rhs, f->u, &f->last_its, &f->last_rnorm, nullptr, nullptr, 0);
    if (rc) return rc;
    TP_TRY(halo_nodes(g, q, f->u, 1));
    TP_LAUNCH(k_pde_node_to_elem, dim3((int)((f->nel + BLK - 1) / BLK)), dim3(BLK), 0, s, q, f->u, out);
    count_launch(g, 8.0 * f->nel + 8.0 * q.owned_nodes(), 8.0 * f->nel);
    return TP_OK;
}

// ---- the three operators of PDEFilt::FilterProject one by one (PDEFilter.cc:198-210), for a host that calls them
// through PETSc names: MatMult(T) / KSPSolve(K_f) / MatMultTranspose(T).  Nodal vectors: local nodes of the grid.
extern "C" int tp_pdefilter_elem_to_node(tp_filter *f, const double *x_elem, double *rhs_nodal) {
    if (!f || f->type != 2 || !x_elem || !rhs_nodal) return TP_ERR_ARG;
    tp_grid *g = f->grid;
    Geom q = f->pde->lv[0].g;
    TP_HIP(hipMemcpyAsync(f->xe, x_elem, sizeof(double) * (size_t)f->nel, hipMemcpyDeviceToDevice, g->stream));
    TP_TRY(exchange_segments(g, f->xe, nullptr, nullptr, f->xe + f->nel, f->lay, 1, f->lay));
    TP_LAUNCH(k_pde_elem_to_node, dim3((int)((q.owned_nodes() + BLK - 1) / BLK)), dim3(BLK), 0, g->stream, q, f->xe, 1.0,
              rhs_nodal, f->rhs);  // vol = 1: T x itself (the caller scales, PDEFilter.cc:202)
    count_launch(g, 8.0 * f->nel + 16.0 * q.owned_nodes(), 9.0 * q.owned_nodes());
    return TP_OK;
}
extern "C" int tp_pdefilter_solve(tp_filter *f, const double *rhs_nodal, double *u_nodal) {
    if (!f || f->type != 2 || !rhs_nodal || !u_nodal) return TP_ERR_ARG;
    return f->pde->solve(rhs_nodal, u_nodal, &f->last_its, &f->last_rnorm, nullptr, nullptr, 0);
}
extern "C" int tp_pdefilter_node_to_elem(tp_filter *f, const double *u_nodal, double *x_elem) {
    if (!f || f->type != 2 || !u_nodal || !x_elem) return TP_ERR_ARG;
    tp_grid *g = f->grid;
    Geom q = f->pde->lv[0].g;
    TP_TRY(halo_nodes(g, q, const_cast<double *>(u_nodal), 1));
    TP_LAUNCH(k_pde_node_to_elem, dim3((int)((f->nel + BLK - 1) / BLK)), dim3(BLK), 0, g->stream, q, u_nodal, x_elem);
    count_launch(g, 8.0 * f->nel + 8.0 * q.owned_nodes(), 8.0 * f->nel);
    return TP_OK;
}
// y = K_f u (the assembled Helmholtz matrix of PDEFilt::MatAssemble): MatMult on it
extern "C" int tp_pdefilter_apply(tp_filter *f, const double *u_nodal, double *y_nodal) {
    if (!f || f->type != 2 || !u_nodal || !y_nodal) return TP_ERR_ARG;
    return f->pde->apply(0, const_cast<double *>(u_nodal), y_nodal);
}

// y = H x, the un-normalised cone filter (MatMult(H, x, y) of Filter.cc:68, :173, :181); types 0 and 1
extern "C" int tp_filter_mult_h(tp_filter *f, const double *x, double *y) {
    if (!f || !x || !y || f->type > 1) return TP_ERR_ARG;
    TP_TRY(filter_fill(f, x, nullptr, 0));
    return filter_conv(f, y, nullptr, nullptr);
}
extern "C" int tp_filter_create(tp_filter **out, tp_grid *g, int filterType, double rmin, const tp_solver_opts *po) {
    if (!out || !g) return TP_ERR_ARG;
    tp_filter *f = new tp_filter();
    f->grid = g;
    f->type = filterType;
    f->R = rmin;
    f->conn = 0;
    f->nel = (long)g->ex * g->ey * g->ez_own;
    f->lay = (long)g->ex * g->ey;
    f->wtab = f->Hs = f->xg = f->tmp = f->d_KF = f->xe = f->rhs = f->u = nullptr;
    f->pde = nullptr;
    f->last_its = 0;
    f->last_rnorm = 0.0;
    f->violations = 0;
    const double dx = g->o.hx, dy = g->o.hy, dz = g->o.hz;
    if (filterType == 0 || filterType == 1) {
        // ElemConn, Filter.cc:326-327
        int conn = (int)fmax(ceil(rmin / dx) - 1, fmax(ceil(rmin / dy) - 1, ceil(rmin / dz) - 1));
        conn = std::min(conn, std::min(g->ex / 2, std::min(g->ey / 2, g->ez_glob / 2)));
        if (conn < 0) conn = 0;
        if (g->nranks > 1 && conn > g->ez_own) {
            delete f;
            return TP_ERR_ARG;
        }
        f->conn = conn;
        const int w1 = 2 * conn + 1;
        std::vector<double> w((size_t)w1 * w1 * w1);
        for (int dk = -conn; dk <= conn; dk++)
            for (int dj = -conn; dj <= conn; dj++)
                for (int di = -conn; di <= conn; di++) {
                    double dist = sqrt((di * dx) * (di * dx) + (dj * dy) * (dj * dy) + (dk * dz) * (dk * dz));
                    w[((size_t)(dk + conn) * w1 + (dj + conn)) * w1 + (di + conn)] = dist < rmin ? rmin - dist : 0.0;  // strict, :430
                }
        TP_HIP(hipMalloc((void **)&f->wtab, sizeof(double) * w.size()));
        TP_HIP(hipMemcpy(f->wtab, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice));
        TP_HIP(hipMalloc((void **)&f->Hs, sizeof(double) * (size_t)f->nel));
        TP_HIP(hipMalloc((void **)&f->tmp, sizeof(double) * (size_t)f->nel));
        const size_t ng = (size_t)(g->ez_own + 2 * conn) * f->lay;
        TP_HIP(hipMalloc((void **)&f->xg, sizeof(double) * ng));
        // Hs = H * 1 (:445-448)
        TP_LAUNCH(k_set, dim3(grid_for((long)ng)), dim3(BLK), 0, g->stream, f->xg, 1.0, (long)ng);
        TP_TRY(filter_conv(f, f->Hs, nullptr, nullptr));
    } else if (filterType == 2) {
        tp_solver_opts o;
        if (po) {
            o = *po;
        } else {
            tp_solver_default_opts(&o);
            o.nlvls = 3;       // PDEFilter.cc:32
            o.rtol = 1.0e-8;   // :280
            o.dtol = 1.0e3;    // :282
            o.max_it = 60;     // :283
            o.nsmooth = 2;
            o.ncoarse = 10;    // :357
        }
        const int fdiv = 1 << (o.nlvls - 1);
        if (g->ex % fdiv || g->ey % fdiv || g->ez_own % fdiv) {
            delete f;
            return TP_ERR_ARG;
        }
        f->elemVol = dx * dy * dz;
        f->KF.resize((size_t)64 * o.nlvls);
        helmholtz_element_box(dx, dy, dz, rmin / 2.0 / sqrt(3), f->KF.data());  // R conversion, :30
        double W[512];
        host_W(W);
        for (int l = 1; l < o.nlvls; l++)  // Galerkin: constant coefficients -> one 8x8 matrix per level
            for (int I = 0; I < 8; I++)
                for (int J = 0; J < 8; J++) {
                    double s = 0.0;
                    for (int c = 0; c < 8; c++)
                        for (int a = 0; a < 8; a++)
                            for (int b = 0; b < 8; b++)
                                s += W[(c * 8 + a) * 8 + I] * f->KF[(size_t)64 * (l - 1) + 8 * a + b] * W[(c * 8 + b) * 8 + J];
                    f->KF[(size_t)64 * l + 8 * I + J] = s;
                }
        TP_HIP(hipMalloc((void **)&f->d_KF, sizeof(double) * f->KF.size()));
        TP_HIP(hipMemcpy(f->d_KF, f->KF.data(), sizeof(double) * f->KF.size(), hipMemcpyHostToDevice));
        {   // the levels' operators as 27-point stencils (operators.h: ScalarStencilOp): one 27 x 27 table per level
            std::vector<double> Wt((size_t)729 * o.nlvls);
            for (int l = 0; l < o.nlvls; l++) pde_stencil_table(f->KF.data() + (size_t)64 * l, Wt.data() + (size_t)729 * l);
            TP_HIP(hipMalloc((void **)&f->d_wtab, sizeof(double) * Wt.size()));
            TP_HIP(hipMemcpy(f->d_wtab, Wt.data(), sizeof(double) * Wt.size(), hipMemcpyHostToDevice));
        }
        f->pde = new MGSolver<1>();
        MGSolver<1> &mg = *f->pde;
        mg.grid = g;
        mg.nlv = o.nlvls;
        mg.opt = o;
        TP_TRY(mg.alloc_levels());
        for (int l = 0; l < mg.nlv; l++) {
            Level<1> &L = mg.lv[l];
            L.kind = LV_MATFREE;
            L.KE = f->d_KF + 64 * l;
            L.wtab = f->d_wtab + 729 * l;
            L.E = nullptr;
            L.mask = nullptr;
            L.S = L.Kel = nullptr;
            TP_TRY(mg.setup_matfree_level(l, f->KF.data()));
        }
        mg.ready = true;
        if (o.ksp_mode != 0 && o.ksp_mode != 1) return TP_ERR_ARG;
        if (o.ksp_mode == 1 && g->has_comm) {
            fprintf(stderr, "topopt_amd: ksp_mode 1 (the reference's FGMRES / GMRES configuration) runs on one device only\n");
            return TP_ERR_ARG;
        }
        if (o.ksp_mode == 0) TP_TRY(mg.estimate_spectra(1));
        Geom q = mg.lv[0].g;
        TP_HIP(hipMalloc((void **)&f->xe, sizeof(double) * (size_t)q.elems_stored()));
        TP_HIP(hipMalloc((void **)&f->rhs, sizeof(double) * (size_t)q.nodes()));
        TP_HIP(hipMalloc((void **)&f->u, sizeof(double) * (size_t)q.nodes()));
        TP_HIP(hipMemset(f->u, 0, sizeof(double) * (size_t)q.nodes()));
        TP_HIP(hipMemset(f->rhs, 0, sizeof(double) * (size_t)q.nodes()));
        TP_HIP(hipMemset(f->xe, 0, sizeof(double) * (size_t)q.elems_stored()));
    }
    *out = f;
    return TP_OK;
}
extern "C" int tp_filter_destroy(tp_filter *f) {
    if (!f) return TP_OK;
    (void)hipStreamSynchronize(f->grid->stream);
    for (double *p : {f->wtab, f->Hs, f->xg, f->tmp, f->d_KF, f->d_wtab, f->xe, f->rhs, f->u}) (void)hipFree(p);
    if (f->pde) {
        f->pde->free_levels();
        delete f->pde;
    }
    delete f;
    return TP_OK;
}
extern "C" int tp_filter_stencil_width(const tp_filter *f) { return f->conn; }
extern "C" int tp_filter_get_kf(const tp_filter *f, double *kf_host_64) {
    if (!f || !kf_host_64 || f->type != 2 || f->KF.size() < 64) return TP_ERR_ARG;
    std::memcpy(kf_host_64, f->KF.data(), sizeof(double) * 64);
    return TP_OK;
}
extern "C" int tp_filter_get_hs(tp_filter *f, double *Hs) {
    if (!f->Hs) return TP_ERR_STATE;
    TP_HIP(hipMemcpyAsync(Hs, f->Hs, sizeof(double) * (size_t)f->nel, hipMemcpyDeviceToDevice, f->grid->stream));
    return TP_OK;
}

extern "C" int tp_filter_project(tp_filter *f, const double *x, double *xTilde, double *xPhys, int proj, double beta,
                                 double eta) {
    tp_grid *g = f->grid;
    hipStream_t s = g->stream;
    const long n = f->nel;
    if (f->type == 1) {  // Filter.cc:66-71
        TP_TRY(filter_fill(f, x, nullptr, 0));
        TP_TRY(filter_conv(f, xTilde, f->Hs, nullptr));
    } else if (f->type == 2) {  // :73-102
        TP_TRY(pde_apply(f, x, xTilde));
        const int nb = grid_for(n, MAX_RED_BLOCKS);
        TP_LAUNCH(k_clamp01, dim3(nb), dim3(BLK), 0, s, xTilde, n, g->partials);
        count_launch(g, 16.0 * n, 0.0);
        TP_TRY(reduce_partials<1>(g, nb, S_TMP));
        double v;
        TP_TRY(read_scal(g, S_TMP, 1, &v));
        f->violations = (long)v;
        if (f->violations && g->rank == 0)
            fprintf(stderr, "BOUND VIOLATION IN PDEFILTER - INCREASE RMIN OR MESH RESOLUTION (%ld elements)\n",
                    f->violations);
    } else {  // :104-107
        if (xTilde != x) TP_HIP(hipMemcpyAsync(xTilde, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    }
    if (proj) {  // :110-114
        TP_LAUNCH(k_heaviside, dim3(grid_for(n)), dim3(BLK), 0, s, xPhys, xTilde, beta, eta, n);
        count_launch(g, 16.0 * n, 10.0 * n);
    } else {
        TP_HIP(hipMemcpyAsync(xPhys, xTilde, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    }
    return TP_OK;
}

static int filter_gradient_one(tp_filter *f, const double *x, double *df) {
    if (f->type == 0) {  // Filter.cc:167-177
        TP_TRY(filter_fill(f, df, x, 2));
        TP_TRY(filter_conv(f, df, f->Hs, x));
    } else if (f->type == 1) {  // :178-192
        TP_TRY(filter_fill(f, df, f->Hs, 1));
        TP_TRY(filter_conv(f, df, nullptr, nullptr));
    } else if (f->type == 2) {  // :193-200
        TP_TRY(pde_apply(f, df, df));
    }
    return TP_OK;
}

extern "C" int tp_filter_gradients(tp_filter *f, const double *x, const double *xTilde, double *dfdx, int m,
                                   double **dgdx, int proj, double beta, double eta) {
    tp_grid *g = f->grid;
    const long n = f->nel;
    if (proj) {  // chain rule of the projection, :125-164
        TP_LAUNCH(k_heaviside_chain, dim3(grid_for(n)), dim3(BLK), 0, g->stream, dfdx, xTilde, beta, eta, n);
        for (int i = 0; i < m; i++)
            TP_LAUNCH(k_heaviside_chain, dim3(grid_for(n)), dim3(BLK), 0, g->stream, dgdx[i], xTilde, beta,
                               eta, n);
        count_launch(g, 24.0 * n * (1 + m), 12.0 * n * (1 + m));
    }
    TP_TRY(filter_gradient_one(f, x, dfdx));
    if (f->type == 0) return TP_OK;  // the sensitivity filter leaves dgdx alone (:167-177)
    for (int i = 0; i < m; i++) TP_TRY(filter_gradient_one(f, x, dgdx[i]));
    return TP_OK;
}

extern "C" int tp_filter_mnd(tp_filter *f, const double *x, double *mnd) {
    tp_grid *g = f->grid;
    const int nb = grid_for(f->nel, MAX_RED_BLOCKS);
    TP_LAUNCH(k_mnd, dim3(nb), dim3(BLK), 0, g->stream, x, f->nel, g->partials);
    count_launch(g, 8.0 * f->nel, 3.0 * f->nel);
    TP_TRY(reduce_partials<1>(g, nb, S_TMP));
    double v;
    TP_TRY(read_scal(g, S_TMP, 1, &v));
    *mnd = v / (double)((long)g->ex * g->ey * g->ez_glob);
    return TP_OK;
}
extern "C" int tp_filter_last_pde_its(const tp_filter *f, int *its, double *rnorm) {
    if (its) *its = f->last_its;
    if (rnorm) *rnorm = f->last_rnorm;
    return TP_OK;
}
