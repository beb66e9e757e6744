for d in 3 5; do for now in "" 1; do for mode in hub random; do echo -n "D=$d noweights=$now "; HCM_SCATTER_D=$d NOW=$now python tools/probes/scatter_case.py $mode 2>/dev/null | grep " ms"; done; done; done
bash tools/probes/pmc_scatter.sh 2>&1 | grep -v "^+" 
