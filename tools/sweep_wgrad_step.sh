#!/bin/bash
# In-step A/B of routing the 72-channel 3x3 layers to the own weight-gradient kernel with a smaller footprint
echo "== baseline (maxc=48)"; python bench.py --steps 30 --warmup 5 --no_cpu_baseline 2>/dev/null | grep "^{" | cut -c79-130
for waves in 4 8; do for lds in 24 32 48; do for want in 256 512; do
  echo "== maxc=80 waves=$waves lds=$lds want_wide=$want"
  HCM_WGRAD_MAXC=80 HCM_WGRAD_WAVES_WIDE=$waves HCM_WGRAD_LDS_KB_WIDE=$lds HCM_WGRAD_WANT_WIDE=$want HCM_WGRAD_WN5=2 python bench.py --steps 30 --warmup 5 --no_cpu_baseline 2>/dev/null | grep "^{" | cut -c79-130
done; done; done
echo "== baseline again"; python bench.py --steps 30 --warmup 5 --no_cpu_baseline 2>/dev/null | grep "^{" | cut -c79-130
