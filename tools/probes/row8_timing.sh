# Phase times inside the row-8 kernels: a DIAGNOSTIC copy of the library (-DHCM_ROW8_TIMING: s_memtime stamps per workgroup),
# built here, loaded by tools/probes/row8_timing.py through HCM_LIB.  usage (on the GPU box): row8_timing.sh [crop] [B]
R=${GRAFT_REPO_ROOT:-/root/repo}
D=/tmp/row8diag
rm -rf $D && mkdir -p $D/hcmoco_amd/csrc $D/include && cp $R/include/*.h $D/include/ && cp $R/hcmoco_amd/csrc/*.hip $R/hcmoco_amd/csrc/*.h $D/hcmoco_amd/csrc/ && cd $D/hcmoco_amd/csrc
for f in *.hip; do
  extra=""; [ $f = pointnet2.hip ] && extra="-ffp-contract=off"; [ $f = scatter.hip ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I$R/include -DHCM_ROW8_TIMING $extra -c $f -o ${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o $D/libhcmoco_hip.so
ls -la $D/libhcmoco_hip.so
HCM_LIB=$D/libhcmoco_hip.so python $R/tools/probes/row8_timing.py ${1:-256} ${2:-32}
