O=gpurun_out/r3_16; mkdir -p $O
for cbl in 8 4 2 1; do for mode in random hub; do echo -n "CBL $cbl "; HCM_SCATTER_CBL=$cbl python tools/probes/scatter_case.py $mode 2>/dev/null | grep " ms"; done; done
