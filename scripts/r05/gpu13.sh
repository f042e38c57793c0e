#!/bin/bash
# two-slot storage: tests, then the C4 SpMV times of both formats on this box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_sparse.py -x -q -m gpu -k "two_slot or both_storage" > gpurun_out/r05/t13.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r05/t13.log
mkdir -p pogs_amd/variants; cp pogs_amd/libpogs_amd.so pogs_amd/variants/libpogs_amd_main.so


