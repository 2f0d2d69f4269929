cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/r04_setup.py 2>&1 | grep set-up
TP_LANCZOS_ONE_THREAD=1 python $R/tools/r04_setup.py 2>&1 | grep set-up
rm -rf /tmp/st && timeout 300 rocprofv3 --kernel-trace -d /tmp/st -- python $R/tools/r04_setup.py 4 > /tmp/st.log 2>&1
python $R/tools/setup_trace3.py $(find /tmp/st -name "*.db" | head -1) | grep "set-up\|stream 2\|stream . :"
