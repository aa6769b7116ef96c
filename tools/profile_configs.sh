#!/bin/bash
# rocprofv3 evidence for the BASELINE configs C2..C5: kernel trace + FETCH_SIZE + MfmaUtil passes of
# steady-state Fock builds, summarised on the box.   usage (GPU box): bash tools/profile_configs.sh <tag>
tag=${1:-r03}
repo=$PWD
out=$repo/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in C2 C3 C3pbe C4 C5; do
  rm -rf /tmp/pk /tmp/pf /tmp/pm
  rocprofv3 --kernel-trace --stats -d /tmp/pk -- python $repo/tools/config_step.py $c 10 > $out/${c}_step.json 2> /tmp/pk.err
  python $repo/tools/rocpd_summary.py $(find /tmp/pk -name '*.db' | head -1) | grep -E "^kernel|dqc::" | head -14 > $out/${c}_kernel_stats.txt 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -- python $repo/tools/config_step.py $c 3 > /dev/null 2> /tmp/pf.err
  python $repo/tools/pmc_summary.py $(find /tmp/pf -name '*.db' | head -1) FETCH_SIZE | grep -E "dqc::" | head -8 > $out/${c}_pmc_FETCH_SIZE.txt 2>&1
  rocprofv3 --kernel-trace --pmc MfmaUtil -d /tmp/pm -- python $repo/tools/config_step.py $c 3 > /dev/null 2> /tmp/pm.err
  python $repo/tools/pmc_summary.py $(find /tmp/pm -name '*.db' | head -1) MfmaUtil | grep -E "dqc::" | head -8 > $out/${c}_pmc_MfmaUtil.txt 2>&1
  tail -1 $out/${c}_step.json
done
