# restart through the reference's own code (TopOpt.cc:386-512: VecLoad of x, xPhys, xo1, xo2, U, L; LinearElasticity's
# -restartFileVecSol): 5 iterations straight against 3 + restart + 2, on 1 and on 2 slab processes
OPTS="-ksp_type cg -mg_levels_ksp_type chebyshev -mg_levels_pc_type jacobi -mg_coarse_ksp_type chebyshev -mg_coarse_pc_type jacobi"
REF=/root/repo/host/_refbuild/topopt_ref
for n in 1 2; do
  rm -rf /tmp/rs$n && mkdir -p /tmp/rs$n/a /tmp/rs$n/b && cd /tmp/rs$n/a
  timeout 120 /root/repo/host/slabrun -n $n --same-device $REF -nx 33 -ny 17 -nz 17 -nlvls 3 -maxItr 5 $OPTS 2>&1 | grep -E "^It" | sed "s/time.*//" | tail -n 2
  cd /tmp/rs$n/b
  timeout 120 /root/repo/host/slabrun -n $n --same-device $REF -nx 33 -ny 17 -nz 17 -nlvls 3 -maxItr 3 $OPTS > /dev/null 2>&1
  f=$(ls -t Restart0?.dat | head -n 1); i=${f%.dat}_itr_f0.dat; s=RestartSol${f#Restart}
  timeout 120 /root/repo/host/slabrun -n $n --same-device $REF -nx 33 -ny 17 -nz 17 -nlvls 3 -maxItr 5 -restart 1 -restartFileVec $f -restartFileItr $i -restartFileVecSol $s $OPTS 2>&1 | grep -E "^It|Successful|NOT FOUND|rror" | sed "s/time.*//" | tail -n 4
done
