#!/bin/bash
# a roof for the derivative-ERI launches of a C5 nuclear gradient -- per class the kernel time (kernel trace) and
# the issue statistics of a PMC pass (VALU instructions, busy / wave cycles, cycles waves spent waiting for any instruction).
# usage (GPU box): bash tools/profile_grad_pmc.sh <tag>  ->  gpurun_out/<tag>_grad_classes_kernel_trace.txt, <tag>_grad_classes_pmc.txt
tag=${1:-r06}
repo=$PWD
out=$repo/gpurun_out
mkdir -p $out
export GRAFT_REPO_ROOT=$repo
bash $repo/tools/profile_grad_classes.sh 60 > $out/${tag}_grad_classes_kernel_trace.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_gp
rocprofv3 --kernel-trace --pmc ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES -d /tmp/prof_gp -- python $repo/tools/gpu_grad_breakdown.py > /dev/null 2> /tmp/gp.err
python $repo/tools/pmc_summary2.py $(find /tmp/prof_gp -name '*.db' | head -1) eri_kernel > $out/${tag}_grad_classes_pmc.txt 2>&1
head -12 $out/${tag}_grad_classes_kernel_trace.txt | cut -c1-160; head -8 $out/${tag}_grad_classes_pmc.txt | cut -c1-220
