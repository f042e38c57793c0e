R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/pogs_amd/libpogs_amd.so /tmp/orig.so
cp $R/pogs_amd/variants/libpogs_amd_dbg.so $R/pogs_amd/libpogs_amd.so
python $R/scripts/tmp/spmv_dbg.py 2>&1 | grep -v Librccl
cp /tmp/orig.so $R/pogs_amd/libpogs_amd.so
