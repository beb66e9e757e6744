set -x
O=gpurun_out/r3_6; mkdir -p $O
for cfg in "" "HCM_TWO_STREAMS=0" "PLAIN=1 HCM_TWO_STREAMS=0" "HCM_CONV_KERNEL=0" "HCM_CONV_STATS=0" "HCM_TWO_STREAMS=0 HCM_CONV_KERNEL=0"; do
  echo "=== $cfg" >> $O/det_forward.txt
  (env $cfg timeout 300 python tools/probes/determinism_forward.py 128 8 2>&1 | grep "run 0") >> $O/det_forward.txt
done
echo "=== 256x256 batch 32 default" >> $O/det_forward.txt
(timeout 300 python tools/probes/determinism_forward.py 256 32 2>&1 | grep "run 0") >> $O/det_forward.txt
for rows in 128 256 512 1024; do for var in 12 3; do
  echo "rows $rows variant $var" >> $O/bank_rows.txt
  (HCM_BANK_ROWS=$rows HCM_BANK_VARIANT=$var python tools/bank_sweep.py worker 1048576 16384 fp32; HCM_BANK_ROWS=$rows HCM_BANK_VARIANT=$var python tools/bank_sweep.py worker 131072 16384 fp32) 2>/dev/null | grep "^{" >> $O/bank_rows.txt
done; done
R=$PWD; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python $R/bench.py --steps 6 --warmup 4 --no_cpu_baseline --no_check > $R/$O/prof_stdout.log 2>&1
python $R/tools/probes/loss_section.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) > $R/$O/loss_section.txt 2>&1
cd $R
cat $O/det_forward.txt; cat $O/bank_rows.txt; grep -v "^+" $O/loss_section.txt; grep -E "head_pool|heads_|branch_grad|sample_branches|pixel_sample|bank_pass|Cijk" $O/loss_section.txt | cut -c1-120
