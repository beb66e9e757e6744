# bf16 bank pass: register-ring depth x occupancy (HCM_BANK_VARIANT: 4 = ring 4 (r03 default), 25 / 26 = ring 5 / 6 held to two
# waves per SIMD, 6 = ring 6 at one wave, 14 / 16 = LDS-DMA ring of 4 / 6 stages) on the HBM-resident cells.
R=${GRAFT_REPO_ROOT:-/root/repo}
for cell in "1048576 16384" "4194304 16384" "4194304 65536" "131072 131072"; do
  for v in 4 25 26 6 14 16; do
    echo -n "n_data K = $cell  variant $v  "
    HCM_BANK_VARIANT=$v python $R/tools/bank_sweep.py worker $cell bf16 2>/dev/null | tail -1
  done
done
