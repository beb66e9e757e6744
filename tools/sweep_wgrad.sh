#!/bin/bash
# Parameter sweep of the wide-layer weight-gradient geometry (run on the GPU box): waves across N, LDS cap, chunk target
for lds in 48 64; do for wn in 1 2; do for want in 64 128 192 256; do
  echo "== lds=$lds wn=$wn want_wide=$want"
  HCM_WGRAD_LDS_KB=$lds HCM_WGRAD_WN5=$wn HCM_WGRAD_WN9=$wn HCM_WGRAD_WANT_WIDE=$want python tools/bench_wgrad.py 2>/dev/null | grep "C=K=72 \|C=K=144 \|C=72 K=144\|C=K=18 \|C=K=36 \|C=36 K=72"
done; done; done
