#!/bin/bash
# Generates golden vectors by RUNNING the reference's own PETSc-free arithmetic
# (Hex8Isoparametric & helpers, LinearElasticity.cc:841-1057; PDEFilterMatrix,
# PDEFilter.cc:472-576) in this container.  The function bodies are piped from
# /root/reference straight into g++ -- no reference source is copied into the
# repository; only the numeric outputs (tests/golden/ref_ke.bin, ref_kf.bin) are
# kept.  The prelude below supplies nothing but the two scalar typedefs and the
# member declarations those bodies need to compile outside PETSc.
# Everything else on the path needs PETSc 3.11 and cannot be run here.
set -euo pipefail
REF=${REF:-/root/reference}
OUT=$(cd "$(dirname "$0")" && pwd)
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
{
cat <<'PRE'
#include <cmath>
#include <cstdio>
#include <cstring>
typedef double PetscScalar; typedef int PetscInt;
struct LinearElasticity {
  PetscInt Hex8Isoparametric(PetscScalar*, PetscScalar*, PetscScalar*, PetscScalar, PetscInt, PetscScalar*);
  PetscScalar Dot(PetscScalar*, PetscScalar*, PetscInt);
  void DifferentiatedShapeFunctions(PetscScalar, PetscScalar, PetscScalar, PetscScalar*, PetscScalar*, PetscScalar*);
  PetscScalar Inverse3M(PetscScalar J[][3], PetscScalar invJ[][3]);
};
struct PDEFilt { void PDEFilterMatrix(PetscScalar, PetscScalar, PetscScalar, PetscScalar, PetscScalar*, PetscScalar*); };
PRE
sed -n '841,1057p' "$REF/LinearElasticity.cc"
sed -n '472,576p'  "$REF/PDEFilter.cc"
cat <<'POST'
int main(int argc, char** argv) {
  // cases: dx dy dz nu   (cube h=1; the C1 and default meshes; an anisotropic box; another nu)
  const double cases[][4] = {{1,1,1,0.3},{1.0/24,1.0/24,1.0/24,0.3},{1.0/32,1.0/32,1.0/32,0.3},
                             {2.0/48,1.0/20,1.0/28,0.3},{0.03125,0.03125,0.03125,0.25},{1.0/64,1.0/64,1.0/64,0.3}};
  const int nc = sizeof(cases)/sizeof(cases[0]);
  LinearElasticity le; PDEFilt pf;
  FILE* f = fopen(argv[1], "wb"); FILE* g = fopen(argv[2], "wb");
  for (int c = 0; c < nc; c++) {
    double dx=cases[c][0], dy=cases[c][1], dz=cases[c][2], nu=cases[c][3];
    double X[8]={0,dx,dx,0,0,dx,dx,0}, Y[8]={0,0,dy,dy,0,0,dy,dy}, Z[8]={0,0,0,0,dz,dz,dz,dz};
    double ke[576]; le.Hex8Isoparametric(X,Y,Z,nu,0,ke);
    fwrite(cases[c], 8, 4, f); fwrite(ke, 8, 576, f);
    // PDE filter: rmin = 2.56*dy and the reference default 0.08
    const double rmins[2] = {2.56*dy, 0.08};
    for (int r = 0; r < 2; r++) {
      double R = rmins[r]/2.0/sqrt(3); double KF[64], TF[8];
      pf.PDEFilterMatrix(dx,dy,dz,R,KF,TF);
      double hdr[4]={dx,dy,dz,rmins[r]}; fwrite(hdr,8,4,g); fwrite(KF,8,64,g); fwrite(TF,8,8,g);
    }
  }
  fclose(f); fclose(g); return 0;
}
POST
} | g++ -O0 -ffp-contract=off -x c++ - -o "$TMP/refgen"
"$TMP/refgen" "$OUT/ref_ke.bin" "$OUT/ref_kf.bin"
ls -l "$OUT"/ref_ke.bin "$OUT"/ref_kf.bin
