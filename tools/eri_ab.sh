#!/bin/bash
# fill parity tests + per-class kernel times of the C5 (and C4 with arg "c4") ERI fill; DQC_ERI_GROUP=0 for the ungrouped tables
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "integral_kernels_vs_oracle or direct_jk_equals_stored or c4_sampled or tile_store_slices or g_shell or c5_fock_density or df_integrals or direct_context or gradient or grad" 2>&1 | tail -5
mkdir -p gpurun_out
bash tools/eri_class_times.sh > gpurun_out/eri_c5_grouped.txt 2>&1; tail -${2:-30} gpurun_out/eri_c5_grouped.txt
if [ "$1" == "c4" ]; then bash tools/eri_class_times_c4.sh > gpurun_out/eri_c4_grouped.txt 2>&1; tail -12 gpurun_out/eri_c4_grouped.txt; fi
