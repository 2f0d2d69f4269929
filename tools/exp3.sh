export TMPDIR=/tmp
for a in 1 2 3 4 5 6; do
echo "ablation $a"; TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_abl$a.so timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
done
echo baseline; timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
bash tools/pmc_sq.sh
