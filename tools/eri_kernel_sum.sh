#!/bin/bash
# GPU time of the ERI class kernels of one 20-atom cc-pVDZ fill (rocprofv3 kernel trace; host-side table building excluded)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_e5
rocprofv3 --kernel-trace --stats -d /tmp/prof_e5 -- python $GRAFT_REPO_ROOT/tools/gpu_eri_c5.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/eri_kernel_sum.py $(find /tmp/prof_e5 -name "*.db" | head -1) 3
