// elements.h -- host-side element matrices of the product (computed once per
// context).  Same formulas and accumulation order as the reference so that the
// matrices agree with it to the last bit (checked against tests/golden/ref_*.bin).
#pragma once
#include <cmath>
#include <cstring>

// 8-node isoparametric hexahedron, 2x2x2 Gauss, unit Young's modulus:
// KE = sum_gp w |J| B^T C B       (LinearElasticity.cc:841-998, redInt = 0)
inline void hex8_stiffness_box(double dx, double dy, double dz, double nu, double *ke /*576*/) {
    const double X[8] = {0.0, dx, dx, 0.0, 0.0, dx, dx, 0.0};
    const double Y[8] = {0.0, 0.0, dy, dy, 0.0, 0.0, dy, dy};
    const double Z[8] = {0.0, 0.0, 0.0, 0.0, dz, dz, dz, dz};
    const double sgx[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sgy[8] = {-1, -1, 1, 1, -1, -1, 1, 1},
                 sgz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
    const double lambda = nu / ((1.0 + nu) * (1.0 - 2.0 * nu)), mu = 1.0 / (2.0 * (1.0 + nu));
    double C[6][6] = {};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) C[i][j] = lambda;
        C[i][i] = lambda + 2.0 * mu;
        C[i + 3][i + 3] = mu;
    }
    // engineering strain rows: xx, yy, zz, xy, yz, zx.  sel[d][row] = displacement
    // component whose d-derivative enters that row (-1: none)
    const int sel[3][6] = {{0, -1, -1, 1, -1, 2}, {-1, 1, -1, 0, 2, -1}, {-1, -1, 2, -1, 1, 0}};
    const double gp[2] = {-0.577350269189626, 0.577350269189626};
    std::memset(ke, 0, sizeof(double) * 576);
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++)
            for (int c = 0; c < 2; c++) {
                const double xi = gp[a], eta = gp[b], zeta = gp[c];
                double dN[3][8];
                for (int n = 0; n < 8; n++) {
                    dN[0][n] = (sgx[n] * 0.125) * (1.0 + sgy[n] * eta) * (1.0 + sgz[n] * zeta);
                    dN[1][n] = (sgy[n] * 0.125) * (1.0 + sgx[n] * xi) * (1.0 + sgz[n] * zeta);
                    dN[2][n] = (sgz[n] * 0.125) * (1.0 + sgx[n] * xi) * (1.0 + sgy[n] * eta);
                }
                double J[3][3];
                for (int r = 0; r < 3; r++) {
                    double sx = 0.0, sy = 0.0, sz = 0.0;
                    for (int n = 0; n < 8; n++) {
                        sx = sx + dN[r][n] * X[n];
                        sy = sy + dN[r][n] * Y[n];
                        sz = sz + dN[r][n] * Z[n];
                    }
                    J[r][0] = sx;
                    J[r][1] = sy;
                    J[r][2] = sz;
                }
                const double det = J[0][0] * (J[1][1] * J[2][2] - J[2][1] * J[1][2]) -
                                   J[0][1] * (J[1][0] * J[2][2] - J[2][0] * J[1][2]) +
                                   J[0][2] * (J[1][0] * J[2][1] - J[2][0] * J[1][1]);
                double iJ[3][3];
                iJ[0][0] = (J[1][1] * J[2][2] - J[2][1] * J[1][2]) / det;
                iJ[0][1] = -(J[0][1] * J[2][2] - J[0][2] * J[2][1]) / det;
                iJ[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
                iJ[1][0] = -(J[1][0] * J[2][2] - J[1][2] * J[2][0]) / det;
                iJ[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
                iJ[1][2] = -(J[0][0] * J[1][2] - J[0][2] * J[1][0]) / det;
                iJ[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
                iJ[2][1] = -(J[0][0] * J[2][1] - J[0][1] * J[2][0]) / det;
                iJ[2][2] = (J[0][0] * J[1][1] - J[1][0] * J[0][1]) / det;
                const double weight = 1.0 * 1.0 * 1.0 * det;
                double B[6][24] = {};
                for (int ll = 0; ll < 3; ll++) {
                    double beta[6][3];
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 3; j++)
                            beta[i][j] = iJ[0][ll] * (sel[0][i] == j ? 1.0 : 0.0) + iJ[1][ll] * (sel[1][i] == j ? 1.0 : 0.0) +
                                         iJ[2][ll] * (sel[2][i] == j ? 1.0 : 0.0);
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 24; j++) B[i][j] = B[i][j] + beta[i][j % 3] * dN[ll][j / 3];
                }
                for (int i = 0; i < 24; i++)
                    for (int j = 0; j < 24; j++)
                        for (int k = 0; k < 6; k++)
                            for (int l = 0; l < 6; l++) ke[j + 24 * i] = ke[j + 24 * i] + weight * (B[k][i] * C[k][l] * B[l][j]);
            }
}

// Helmholtz filter element matrix KF = R^2 int grad N . grad N + int N N on a
// box element, closed form (PDEFilter.cc:472-565); entries depend only on
// which axes the two nodes differ along.
inline void helmholtz_element_box(double dx, double dy, double dz, double RR, double *KF /*64*/) {
    const double pre1 = 1.0 / dx / dy, pre2 = 1 / dz;
    const double r2 = RR * RR, x2 = dx * dx, y2 = dy * dy, z2 = dz * dz;
    const double rx = r2 * x2;
    const double gxy = rx * y2, gxz = rx * z2, gyz = r2 * y2 * z2, mass = x2 * y2 * z2;
    const double a3 = 3.0 * gxy, b3 = 3.0 * gxz, c3 = 3.0 * gyz, a6 = 6.0 * gxy, b6 = 6.0 * gxz, c6 = 6.0 * gyz;
    double v[8];  // index = dx_differs + 2*dy_differs + 4*dz_differs
    v[0] = pre1 * pre2 * (a3 + b3 + c3 + mass) / 27.0;
    v[1] = pre1 * pre2 * (a3 + b3 - c6 + mass) / 54.0;
    v[3] = pre1 * pre2 * (a3 - b6 - c6 + mass) / 108.0;
    v[2] = pre1 * pre2 * (a3 - b6 + c3 + mass) / 54.0;
    v[4] = -(pre1 * pre2 * (a6 - b3 - c3 - mass) / 54.0);
    v[5] = -(pre1 * pre2 * (a6 - b3 + c6 - mass) / 108.0);
    v[7] = -(pre1 * pre2 * (a6 + b6 + c6 - mass) / 216.0);
    v[6] = -(pre1 * pre2 * (a6 + b6 - c3 - mass) / 108.0);
    const int lx[8] = {0, 1, 1, 0, 0, 1, 1, 0}, ly[8] = {0, 0, 1, 1, 0, 0, 1, 1}, lz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    for (int p = 0; p < 8; p++)
        for (int q = 0; q < 8; q++)
            KF[8 * p + q] = v[(lx[p] != lx[q]) + 2 * (ly[p] != ly[q]) + 4 * (lz[p] != lz[q])];
}
