# bf16 bank pass: kernel x register-ring depth (HCM_BANK_VARIANT: 34 / 35 = csrc/bank_lean.hip ring 4 (default) / 5, 26 = general kernel ring 6 at two
# waves per SIMD (default before bank_lean), 4 = general ring 4 (r03 default), 14 = LDS-DMA ring of 4 stages) on the HBM-resident cells.
R=${GRAFT_REPO_ROOT:-/root/repo}
for cell in "1048576 16384" "4194304 16384" "4194304 65536" "131072 131072"; do
  for v in 34 35 26 4 14; do
    echo -n "n_data K = $cell  variant $v  "
    HCM_BANK_VARIANT=$v python $R/tools/bank_sweep.py worker $cell bf16 2>/dev/null | tail -1
  done
done
