#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_k3 -o t -- python $R/scripts/bench_kernels.py --shapes 4096x4096 --no-balance > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace -d $O/pmc_k3 -o p -- python $R/scripts/bench_kernels.py --shapes 4096x4096 --no-balance > /dev/null 2>&1
cd $R; python scripts/rocpd_summary.py $O/prof_k3/t_results.db $O/pmc_k3/p_results.db | grep -E "ortho|kernel  " | cut -c1-190
