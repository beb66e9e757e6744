#!/bin/bash
# The measurement set behind profiles/rNN_final_* (run on the GPU box from the repo root):
#   bash tools/final_profiles.sh r02      -> gpurun_out/final/r02_*
set -e
TAG=${1:-rXX}
R=$PWD
OUT=$R/gpurun_out/final
mkdir -p $OUT
python bench.py --steps 30 --warmup 5 2>/dev/null < /dev/null | grep "^{" | tail -1 > $OUT/${TAG}_final_bench_line.json
HCM_FORCE_COLLECTIVES=1 python bench.py --steps 30 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep "^{" | tail -1 > $OUT/${TAG}_forced_collectives_bench_line.json
: > $OUT/${TAG}_secondary_configs.log
for flags in "--width 32" "--arch HRNetPN" "--arch HRNetPN --width 32" "--bank_dtype bf16 --fmap_dtype bf16 --nce_k 131072" "--bank_dtype bf16 --fmap_dtype bf16 --nce_k 131072 --encoder_dtype bf16" "--nce_k 65536"; do
  echo "== $flags" >> $OUT/${TAG}_secondary_configs.log
  python bench.py --steps 20 --warmup 5 --no_cpu_baseline $flags 2>/dev/null < /dev/null | grep "^{" | tail -1 >> $OUT/${TAG}_secondary_configs.log
done
# the reference's recipe of record (scripts/SecondStage/train_ntumpiirgbd2s_hrnet_w18.sh: 320 x 320, 56 per GPU, MPII-16)
: > $OUT/${TAG}_recipe_320_b56_mpii16.log
for flags in "--size 320 --batch_per_gpu 56 --skeleton mpii" "--size 320 --batch_per_gpu 56 --skeleton mpii --arch HRNetPN"; do
  echo "== $flags" >> $OUT/${TAG}_recipe_320_b56_mpii16.log
  python bench.py --steps 12 --warmup 4 --no_cpu_baseline $flags 2>/dev/null < /dev/null | grep "^{" | tail -1 >> $OUT/${TAG}_recipe_320_b56_mpii16.log
done
HCM_DETERMINISTIC=1 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null < /dev/null | grep "^{" | tail -1 > $OUT/${TAG}_deterministic_mode_bench_line.json
python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no_cpu_baseline --no_check 2>/dev/null < /dev/null | grep "^{" | tail -1 > $OUT/${TAG}_two_ranks_one_gpu_gloo_bench_line.json || true
(python tools/bench_pointnet2.py 2>&1 | grep -v amdgpu.ids) > $OUT/${TAG}_pointnet2_ops_config4.txt
(python tools/probes/phase_times.py 2>&1 | tail -12) > $OUT/${TAG}_phase_times.txt || true
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python $R/bench.py --steps 10 --warmup 5 --no_cpu_baseline > $OUT/${TAG}_final_bench_stdout_profiled.log 2>/dev/null
cp $(find /tmp/fp -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_final_bench_kernel_stats.csv
python $R/tools/step_profile.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) $OUT/${TAG}_final_bench_one_step_summary.csv > /dev/null
python $R/tools/timeline.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_final_bench_timeline.txt 2>&1 || true
python $R/tools/probes/loss_section.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_loss_section.txt 2>&1 || true
bash $R/tools/probes/pmc_strip_mfma.sh $TAG > $OUT/${TAG}_strip_mfma_pmc.log 2>&1 < /dev/null || true
cp $R/gpurun_out/${TAG}_strip_mfma_pmc_*.json $OUT/ 2>/dev/null || true
bash $R/tools/probes/hrnetpn_timeline.sh 18 > $OUT/${TAG}_hrnetpn_timeline.txt 2>&1 < /dev/null || true
(ARCH=HRNetPN python $R/tools/probes/phase_times.py 2>&1 | tail -10) > $OUT/${TAG}_phase_times_hrnetpn.txt || true
ls -la $OUT
