# builds ablation variants of the fine kernel as separate libraries (timing only: results are WRONG by construction)
cd $(dirname $0)/../topopt_in_petsc_amd/csrc
for a in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=default -Wno-unused-function -DFT_ABL=$a -shared -o ../libtopopt_abl$a.so topopt_amd.hip &
done
wait
