# the round's final check on the GPU box (repo root): full GPU suite, smoke, then the measurement set behind profiles/r06_*
O=gpurun_out/r6_final; mkdir -p $O
(timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 2>&1 | tail -30) > $O/pytest_full.log 2>&1
tail -3 $O/pytest_full.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok")
timeout 2700 bash tools/final_profiles.sh r06 > $O/final_profiles.log 2>&1
tail -5 $O/final_profiles.log
