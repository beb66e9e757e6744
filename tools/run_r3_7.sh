set -x
O=gpurun_out/r3_7; mkdir -p $O
(timeout 1500 python -m pytest tests/test_section_gpu.py tests/test_fmap_gpu.py tests/test_trace_moco.py tests/test_trace.py -x -q -m gpu 2>&1 | tail -15) > $O/pytest.log 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1) > $O/bench_default.json
(HCM_BANK_VARIANT=3 HCM_BANK_ROWS=512 timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1) > $O/bench_reg3_512.json
(HCM_BANK_VARIANT=3 timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1) > $O/bench_reg3_256.json
(HCM_BANK_VARIANT=12 HCM_BANK_ROWS=128 timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1) > $O/bench_glds2_128.json
python tools/probes/phase_times.py > $O/phase_times.txt 2>&1 || true
bash tools/probes/pmc_strip.sh > $O/strip_sq_counters.txt 2>&1
R=$PWD; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python $R/bench.py --steps 6 --warmup 4 --no_cpu_baseline --no_check > $R/$O/prof_stdout.log 2>&1
python $R/tools/probes/loss_section.py $(find /tmp/fp -name "*kernel_trace.csv" | head -1) > $R/$O/loss_section.txt 2>&1
cd $R
(DET=1 timeout 900 python tools/probes/determinism_grads.py 256 32 2>&1 | grep -v "^\[\|Gloo\|^ *$" | tail -34) > $O/det_grads_256.txt
(DET=0 timeout 900 python tools/probes/determinism_grads.py 256 32 2>&1 | grep -v "^\[\|Gloo\|^ *$" | tail -8) > $O/det_grads_256_nodet.txt
tail -n 5 $O/pytest.log; for f in bench_default bench_reg3_512 bench_reg3_256 bench_glds2_128; do python -c "
import json
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('checked')); print([ (s['kernel'][:24], s['avg_launch_ms']) for s in d['roofline_secondary']])"; done; tail -9 $O/phase_times.txt; cat $O/strip_sq_counters.txt | tail -6; grep -v "^+" $O/loss_section.txt; grep -E "branch_grad|strip_kernel" $O/loss_section.txt | cut -c1-100; cat $O/det_grads_256.txt; cat $O/det_grads_256_nodet.txt
