# slab geometries at the edge (each run bounded to 60 s): two levels keep the coarsest level distributed; one element layer
# per rank on it is refused at creation, on every rank alike
for cfg in "2 16 8 8 2" "4 16 8 8 2" "4 16 8 16 3"; do
set -- $cfg
echo "== ranks $1 mesh $2 $3 $4 levels $5"
timeout 60 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29511 tests/mp_gloo_worker.py gpu $2 $3 $4 $5 2>&1 | grep -E "gpu OK|TopOptError|exitcode|Assertion|assert |topopt_amd" | head -6
done
