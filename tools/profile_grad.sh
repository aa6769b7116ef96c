#!/bin/bash
# where does a nuclear gradient go?  kernel trace of tools/gpu_grad_time.py (C5: PBE, LDA, RHF)
repo=$PWD; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_g
python $repo/tools/gpu_grad_time.py 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats -d /tmp/prof_g -- python $repo/tools/gpu_grad_time.py > /dev/null 2>&1
python $repo/tools/rocpd_summary.py $(find /tmp/prof_g -name "*.db" | head -1) | head -${1:-22} | cut -c1-70,88-150
