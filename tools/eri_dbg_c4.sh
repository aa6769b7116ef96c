#!/bin/bash
# timing experiments on the C4 fill (DQC_ERI_DBG bits: 1 no primitive loops, 2 no output phase, 4 no tile stores)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for d in ${@:-0 1 2 3}; do
  export DQC_ERI_DBG=$d
  bash tools/eri_class_times_c4.sh > gpurun_out/eri_c4_dbg$d.txt 2>&1
  echo "== DQC_ERI_DBG=$d"; cut -c18-52,60-100 gpurun_out/eri_c4_dbg$d.txt | tail -14
done
