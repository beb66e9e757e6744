# Phase times inside strip_kernel<Dense> (stats and grad pass): a DIAGNOSTIC copy of the library (-DHCM_STRIP_TIMING:
# s_memtime stamps of wave 0 of every workgroup, summed per phase of the key-tile loop), built here and loaded through HCM_LIB.
# usage (GPU box): strip_timing.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
D=/tmp/stripdiag
rm -rf $D && mkdir -p $D/hcmoco_amd/csrc $D/include && cp $R/include/*.h $D/include/ && cp $R/hcmoco_amd/csrc/*.hip $R/hcmoco_amd/csrc/*.h $D/hcmoco_amd/csrc/ && cd $D/hcmoco_amd/csrc
for f in *.hip; do
  extra=""; [ $f = pointnet2.hip ] && extra="-ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None"; [ $f = scatter.hip ] && extra="-fno-slp-vectorize"; [ $f = bank_lean.hip ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I$R/include -DHCM_STRIP_TIMING $extra -c $f -o ${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o $D/libhcmoco_hip.so
HCM_LIB=$D/libhcmoco_hip.so python $R/tools/probes/strip_timing.py
