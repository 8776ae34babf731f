#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
{ ORDER=ref timeout 300 python scripts/profile_phases.py 3 4096; } > $O/r04_wave_time3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --output-format csv -d $O/r04_pcs_ref -- python $R/scripts/ref_order_time.py 3 4096 > $O/r04_pcs_ref.log 2>&1; echo "pcs stochastic rc=$?"
ls -la $O/r04_pcs_ref/* | head
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100 --output-format csv -d $O/r04_pcs_ref_ht -- python $R/scripts/ref_order_time.py 3 4096 > $O/r04_pcs_ref_ht.log 2>&1; echo "pcs host_trap rc=$?"
ls -la $O/r04_pcs_ref_ht/* | head
