export TMPDIR=/tmp
rm -rf gpurun_out/pmc_sq
i=0
for set in "SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_SMEM"; do
i=$((i+1))
timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_sq/s$i -- python tools/pmc_traffic.py 128 128 128 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "k_matfree_tile" in k or "k_fine_tile" in k or "k_scale" in k:
            acc[k[:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
