# slab geometries at the edge (each run bounded): odd rank counts, one coarsest-level layer per rank (replicated level),
# two levels (distributed coarsest level), the refusal of one layer per rank on a distributed level
for cfg in "3 16 8 24 3" "8 16 8 64 4" "2 32 16 16 4" "2 16 8 8 2" "4 16 8 8 2"; do
set -- $cfg
echo "== ranks $1 mesh $2 $3 $4 levels $5"
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29511 tests/mp_gloo_worker.py gpu $2 $3 $4 $5 2>&1 | grep -E "gpu OK|TopOptError|exitcode|Assertion|assert |topopt_amd" | head -9
done
