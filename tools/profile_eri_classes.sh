#!/bin/bash
# per-class evidence for the one-off ERI fill: kernel-trace times of a 20-atom cc-pVDZ and the naphthalene /
# cc-pVTZ fill, and a PMC pass (VALU activity, waits, LDS / VMEM instruction counts) of the former.
# usage (GPU box): bash tools/profile_eri_classes.sh <tag>   ->  gpurun_out/<tag>/eri_*.txt
tag=${1:-r03}
repo=$PWD
out=$repo/gpurun_out/$tag
mkdir -p $out
export GRAFT_REPO_ROOT=$repo
bash $repo/tools/eri_class_times.sh > $out/eri_classes_c5_kernel_trace.txt 2>&1
bash $repo/tools/eri_class_times_c4.sh 60 > $out/eri_classes_c4_kernel_trace.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ep
rocprofv3 --kernel-trace --pmc ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES -d /tmp/prof_ep -- python $repo/tools/gpu_eri_c5.py > /dev/null 2> /tmp/ep.err
python $repo/tools/pmc_summary2.py $(find /tmp/prof_ep -name '*.db' | head -1) eri_kernel > $out/eri_classes_c5_pmc.txt 2>&1
tail -1 $out/eri_classes_c5_kernel_trace.txt; tail -1 $out/eri_classes_c4_kernel_trace.txt; head -5 $out/eri_classes_c5_pmc.txt | cut -c1-220
