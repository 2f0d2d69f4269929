# instruction mix / wait reasons / LDS conflicts of the level-1 operator: separate --pmc passes (no tracing domains)
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_l1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
i=$((i+1))
timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_l1/s$i -- python tools/pmc_level1.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("gpurun_out/pmc_l1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "k_matfree_tile" in k and ", 1>" in k:
            acc[k[:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/r02_pmc_counters_level1.json", "w"), indent=1)
PY
rm -rf gpurun_out/pmc_l1
