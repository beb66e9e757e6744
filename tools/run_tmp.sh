R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python $R/bench.py --steps 8 --warmup 4 --no_cpu_baseline --no_check --arch HRNetPN > /tmp/o.log 2>/dev/null
cd $R
mkdir -p gpurun_out/r3_25
python tools/step_profile.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) gpurun_out/r3_25/hrnetpn_one_step_summary.csv > /dev/null
python tools/timeline.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) > gpurun_out/r3_25/hrnetpn_timeline.txt 2>&1
head -60 gpurun_out/r3_25/hrnetpn_one_step_summary.csv | cut -c1-170
head -12 gpurun_out/r3_25/hrnetpn_timeline.txt
