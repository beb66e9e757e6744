set -x
O=gpurun_out/r3_3; mkdir -p $O
(timeout 1200 python -m pytest tests/test_section_gpu.py -x -q -m gpu 2>&1 | tail -30) > $O/pytest_section.log 2>&1
(timeout 900 python -m pytest tests/test_whole_step_gpu.py -x -q -m gpu -k "config2_stage2 or config5" 2>&1 | tail -8) > $O/pytest_step.log 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1) > $O/bench_fused.json
(HCM_FUSED_SECTION=0 timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1) > $O/bench_unfused.json
(timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1) > $O/bench_fused2.json
(HCM_FUSED_SECTION=0 timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1) > $O/bench_unfused2.json
python tools/probes/phase_times.py > $O/phase_times.txt 2>&1 || true
R=$PWD; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python $R/bench.py --steps 6 --warmup 4 --no_cpu_baseline --no_check > $R/$O/prof_stdout.log 2>&1
python $R/tools/probes/loss_section.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) > $R/$O/loss_section.txt 2>&1
cd $R
timeout 300 tools/probes/gather_ceiling $O/gather_ceiling.json > $O/gather_ceiling.txt 2>&1
tail -n 12 $O/pytest_section.log; tail -n 4 $O/pytest_step.log; for f in bench_fused bench_unfused bench_fused2 bench_unfused2; do head -c 160 $O/$f.json | cut -c60-160; echo; done; tail -12 $O/phase_times.txt; grep -v "^+" $O/loss_section.txt; grep -E "head_pool|heads_|branch_grad|sample_branches|pixel_sample|section_total|Cijk" $O/loss_section.txt | cut -c1-120; cat $O/gather_ceiling.txt
