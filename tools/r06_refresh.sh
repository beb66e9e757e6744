# after the last kernel change of the round: full GPU suite, smoke, the bench line and the rocprofv3-derived files of the set
O=gpurun_out/r6_final; mkdir -p $O gpurun_out/final; R=$PWD; OUT=$R/gpurun_out/final; TAG=r06
(timeout 1200 python -m pytest tests -q -m gpu -x --durations=12 -s 2>&1 | grep "worst per-parameter\|passed\|failed\|FAILED\|^[0-9.]*s call" | tail -22) > $O/pytest_full.log 2>&1
tail -2 $O/pytest_full.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok")
python bench.py --steps 30 --warmup 5 2>/dev/null < /dev/null | grep "^{" | tail -1 > $OUT/${TAG}_final_bench_line.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python $R/bench.py --steps 10 --warmup 5 --no_cpu_baseline > $OUT/${TAG}_final_bench_stdout_profiled.log 2>/dev/null
cp $(find /tmp/fp -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_final_bench_kernel_stats.csv
python $R/tools/step_profile.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) $OUT/${TAG}_final_bench_one_step_summary.csv > /dev/null
python $R/tools/timeline.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_final_bench_timeline.txt 2>&1 || true
python $R/tools/probes/loss_section.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_loss_section.txt 2>&1 || true
(python $R/tools/probes/phase_times.py 2>&1 | tail -12) > $OUT/${TAG}_phase_times.txt || true
tail -3 $OUT/${TAG}_loss_section.txt
