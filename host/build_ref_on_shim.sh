#!/bin/bash
# Build container only: compiles the reference's LinearElasticity.cc / Filter.cc / PDEFilter.cc IN PLACE (unchanged,
# from /root/reference) against include/petsc_compat and links them, with host/ref_driver.cc, against the shim.
# Output: host/_refbuild/ref_on_shim and host/_refbuild/topopt_ref (the reference's whole program), git-ignored; like
# every built artefact they travel to the GPU box.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${REFERENCE_DIR:-/root/reference}
[ -d "$REF" ] || { echo "no reference sources at $REF"; exit 3; }
OUT=$HERE/_refbuild
TMP=$(mktemp -d)
mkdir -p "$OUT"
make -s -C "$HERE" ../topopt_in_petsc_amd/libtopopt_petsc_shim.so
for f in LinearElasticity Filter PDEFilter; do
  g++ -std=c++11 -O2 -w -I"$HERE/../include/petsc_compat" -I"$REF" -c "$REF/$f.cc" -o "$TMP/$f.o"
done
g++ -std=c++11 -O2 -w -I"$HERE/../include/petsc_compat" -I"$REF" -c "$HERE/ref_driver.cc" -o "$TMP/driver.o"
g++ -o "$OUT/ref_on_shim" "$TMP"/driver.o "$TMP"/LinearElasticity.o "$TMP"/Filter.o "$TMP"/PDEFilter.o \
    -L"$HERE/../topopt_in_petsc_amd" -ltopopt_petsc_shim -ltopopt_amd \
    -Wl,-rpath,'$ORIGIN/../../topopt_in_petsc_amd' -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
# the WHOLE reference program, all eight sources unchanged (main.cc, TopOpt.cc, MMA.cc, MPIIO.cc + the three above)
for f in main TopOpt MMA MPIIO; do
  g++ -std=c++11 -O2 -w -I"$HERE/../include/petsc_compat" -I"$REF" -c "$REF/$f.cc" -o "$TMP/$f.o"
done
g++ -o "$OUT/topopt_ref" "$TMP"/main.o "$TMP"/TopOpt.o "$TMP"/MMA.o "$TMP"/MPIIO.o "$TMP"/LinearElasticity.o "$TMP"/Filter.o "$TMP"/PDEFilter.o \
    -L"$HERE/../topopt_in_petsc_amd" -ltopopt_petsc_shim -ltopopt_amd \
    -Wl,-rpath,'$ORIGIN/../../topopt_in_petsc_amd' -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
# the reference's MMA class alone, around host/ref_mma_driver.cc (m > 1 constraints)
g++ -std=c++11 -O2 -w -I"$HERE/../include/petsc_compat" -I"$REF" -c "$HERE/ref_mma_driver.cc" -o "$TMP/mma_driver.o"
g++ -o "$OUT/ref_mma" "$TMP"/mma_driver.o "$TMP"/MMA.o \
    -L"$HERE/../topopt_in_petsc_amd" -ltopopt_petsc_shim -ltopopt_amd \
    -Wl,-rpath,'$ORIGIN/../../topopt_in_petsc_amd' -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
rm -rf "$TMP"
echo "built $OUT/ref_on_shim $OUT/topopt_ref $OUT/ref_mma"
