#!/bin/bash
# C4 (naphthalene / cc-pVTZ) Vxc rectangles kernel: launch duration and HBM traffic (FETCH_SIZE) against the block-count target
repo=$PWD; out=$repo/gpurun_out/${1:-r04q}; mkdir -p $out; cd /tmp && export TMPDIR=/tmp
for b in 512 768 1024 1536 3072; do
  export DQC_VXC_BLOCKS=$b
  rm -rf /tmp/pk /tmp/pf
  rocprofv3 --kernel-trace --stats -d /tmp/pk -- python $repo/tools/config_step.py C4 10 > /dev/null 2> /tmp/pk.err
  t=$(python $repo/tools/rocpd_summary.py $(find /tmp/pk -name '*.db' | head -1) | grep "vxc_ws2" | head -1 | awk '{print $(NF-3)}')
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -- python $repo/tools/config_step.py C4 3 > /dev/null 2> /tmp/pf.err
  f=$(python $repo/tools/pmc_summary.py $(find /tmp/pf -name '*.db' | head -1) FETCH_SIZE | grep "vxc_ws2" | head -1)
  echo "blocks $b  avg_us $t  $f" | tee -a $out/c4_blocks.txt
done
