set -x
O=gpurun_out/r3_5; mkdir -p $O
(timeout 1200 python -m pytest tests/test_section_gpu.py tests/test_bank_gpu.py tests/test_memory_module_gpu.py -x -q -m gpu 2>&1 | tail -30) > $O/pytest.log 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1) > $O/bench_fused.json
(HCM_FUSED_SECTION=0 timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1) > $O/bench_unfused.json
(HCM_BANK_VARIANT=3 timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1) > $O/bench_fused_reg3.json
python tools/probes/phase_times.py > $O/phase_times.txt 2>&1 || true
R=$PWD; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python $R/bench.py --steps 6 --warmup 4 --no_cpu_baseline --no_check > $R/$O/prof_stdout.log 2>&1
python $R/tools/probes/loss_section.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) > $R/$O/loss_section.txt 2>&1
cd $R
timeout 900 python tools/probes/determinism_grads.py 128 8 > $O/det_grads.txt 2>&1
timeout 1500 python tools/bank_sweep.py time $O/bank_time.json > $O/bank_time.log 2>&1
tail -n 6 $O/pytest.log; for f in bench_fused bench_unfused bench_fused_reg3; do python -c "
import json
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('checked'))"; done; tail -9 $O/phase_times.txt; grep -v "^+" $O/loss_section.txt; grep -E "head_pool|heads_|branch_grad|sample_branches|pixel_sample|bank_pass|Cijk" $O/loss_section.txt | cut -c1-120; grep -v "^\[\|Gloo\|^ *$" $O/det_grads.txt | tail -40; cat $O/bank_time.log | cut -c1-250
