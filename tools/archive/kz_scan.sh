export TMPDIR=/tmp
var=$1; shift
for kz in $@; do
env $var=$kz timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kz$kz -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find gpurun_out/prof_kz$kz -name "*.db" | head -n 1) > gpurun_out/kz$kz.csv
rm -rf gpurun_out/prof_kz$kz
echo "$var=$kz"; grep "tile<[02], .>" gpurun_out/kz$kz.csv | cut -c1-100
done
