set -x
O=gpurun_out/r3_2; mkdir -p $O
(timeout 1200 python -m pytest tests/test_section_gpu.py -x -q -m gpu 2>&1 | tail -40) > $O/pytest_section.log 2>&1
(timeout 1500 python -m pytest tests/test_whole_step_gpu.py tests/test_trainer_gpu.py tests/test_trace.py tests/test_memory_module_gpu.py tests/test_bank_gpu.py -x -q -m gpu 2>&1 | tail -40) > $O/pytest_step.log 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>$O/bench.err | tail -3) > $O/bench.json
(HCM_FUSED_SECTION=0 timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1) > $O/bench_unfused.json
(timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1) > $O/bench_fused2.json
R=$PWD; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python $R/bench.py --steps 6 --warmup 4 --no_cpu_baseline --no_check > $R/$O/prof_stdout.log 2>&1
python $R/tools/probes/loss_section.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) > $R/$O/loss_section.txt 2>&1
python $R/tools/step_profile.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) $R/$O/one_step_summary.csv > $R/$O/step_profile.txt 2>&1
cd $R; python tools/probes/phase_times.py > $O/phase_times.txt 2>&1 || true
tail -n 12 $O/pytest_section.log; tail -n 8 $O/pytest_step.log; head -c 600 $O/bench.json; echo; head -c 300 $O/bench_unfused.json; echo; head -c 300 $O/bench_fused2.json; echo; head -3 $O/loss_section.txt
