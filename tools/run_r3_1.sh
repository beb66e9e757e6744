set -x
mkdir -p gpurun_out/r3_1
(timeout 1500 python -m pytest tests/test_whole_step_gpu.py tests/test_bench_launch_gpu.py -x -q -m gpu 2>&1 | tail -40) > gpurun_out/r3_1/pytest_new.log 2>&1
(timeout 900 python -m pytest tests/test_rccl_gpu.py tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r3_1/pytest_rccl.log 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/r3_1/bench.err | tail -3) > gpurun_out/r3_1/bench.json
R=$PWD; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python $R/bench.py --steps 6 --warmup 4 --no_cpu_baseline --no_check > $R/gpurun_out/r3_1/prof_stdout.log 2>&1
python $R/tools/probes/loss_section.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/r3_1/loss_section.txt 2>&1
python $R/tools/step_profile.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) $R/gpurun_out/r3_1/one_step_summary.csv > $R/gpurun_out/r3_1/step_profile.txt 2>&1
cd $R; python tools/probes/phase_times.py > gpurun_out/r3_1/phase_times.txt 2>&1 || true
tail -5 gpurun_out/r3_1/pytest_new.log gpurun_out/r3_1/pytest_rccl.log; head -c 1500 gpurun_out/r3_1/bench.json; head -3 gpurun_out/r3_1/loss_section.txt
