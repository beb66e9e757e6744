O=gpurun_out/r6_b; mkdir -p $O
export SHAPES="16,32,4096,16;32,64,4096,32;64,128,1024,16;64,128,1024,32;128,256,256,16;128,256,256,32;256,512,64,16;256,512,64,32"
for a in 0 1; do echo "== ARITH=$a"; ARITH=$a python tools/bench_conv1x1.py 2>&1 | grep -v amdgpu.ids | cut -c1-125; done | tee $O/conv1x1_v2h.txt
