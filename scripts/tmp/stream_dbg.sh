R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/pogs_amd/libpogs_amd.so /tmp/orig.so
cp $R/pogs_amd/variants/libpogs_amd_sdbg.so $R/pogs_amd/libpogs_amd.so
python $R/scripts/tmp/stream_dbg.py 2>&1 | grep -v -i "librccl\|rocm version\|hostname"
cp /tmp/orig.so $R/pogs_amd/libpogs_amd.so
