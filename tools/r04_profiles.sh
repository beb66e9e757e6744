#!/bin/bash
# Everything behind profiles/r04_* that is not the bank sweep (tools/run_bank_sweep.sh) -- run on the GPU box from the repo root:
#   bash tools/r04_profiles.sh      -> gpurun_out/final/r04_*
R=$PWD
OUT=$R/gpurun_out/final
mkdir -p $OUT
bash tools/final_profiles.sh r04 > $OUT/r04_final_profiles.log 2>&1
(python tools/probes/row8_probe.py 256 32 20 2>&1 | grep -v amdgpu.ids) > $OUT/r04_row8_probe.txt
(python tools/probes/row8_probe.py 320 56 20 2>&1 | grep -v amdgpu.ids) > $OUT/r04_row8_probe_320_b56.txt
(bash tools/probes/row8_timing.sh 256 32 2>&1 | grep -v "amdgpu.ids\|^-rwx") > $OUT/r04_row8_phase_stamps.txt
(SKIP_TIMING=1 bash tools/probes/pmc_row8.sh project_rows_kernel,branch_grad_t_kernel,proj_dw_partial_kernel,stencil_plan_kernel 2>&1 | grep -v amdgpu.ids) > $OUT/r04_row8_counters.txt
(ARCH=HRNetPN python tools/probes/phase_times.py 2>&1 | tail -10) > $OUT/r04_phase_times_hrnetpn.txt
(python tools/probes/hrnetpn_streams.py 2>&1 | tail -1) > $OUT/r04_hrnetpn_streams.txt
ls -la $OUT
