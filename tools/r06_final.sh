# this round's last check on the GPU box (repo root): full GPU suite, smoke, the measurement set behind profiles/r06_*,
# and the row-8 counters after the rows-first / reordered backward and the multiply-by-magic forward
O=gpurun_out/r6_final; mkdir -p $O
(timeout 1200 python -m pytest tests -q -m gpu -x --durations=12 2>&1 | tail -30) > $O/pytest_full.log 2>&1
tail -3 $O/pytest_full.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok")
timeout 1800 bash tools/final_profiles.sh r06 > $O/final_profiles.log 2>&1
tail -5 $O/final_profiles.log
(SKIP_TIMING= timeout 600 bash tools/probes/pmc_row8.sh project_rows_kernel,branch_grad_t_kernel,finest_rows_kernel 2>&1 | grep -v amdgpu.ids) > $O/r06_row8_counters.txt
tail -12 $O/r06_row8_counters.txt
