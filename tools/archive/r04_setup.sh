cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/r04_setup.py 2>&1 | grep set-up
rm -rf /tmp/st && timeout 300 rocprofv3 --kernel-trace -d /tmp/st -- python $R/tools/r04_setup.py 4 > /tmp/st.log 2>&1
python $R/tools/setup_trace3.py $(find /tmp/st -name "*.db" | head -1) | grep -v "k_multi\|k_dia_row_split<3, 0\|k_matfree_tile<0\|k_lanczos\|k_reduce\|k_pw" | head -40
