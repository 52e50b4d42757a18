python -m pytest tests/test_kernels_gpu.py -x -q -k "x3 or sampled or accumulate" > gpurun_out/t_fx.log 2>&1; tail -3 gpurun_out/t_fx.log
python tools/conv_bench.py --iters 20 2>/dev/null > gpurun_out/cb_a.txt; cat gpurun_out/cb_a.txt
P3D_FX_MIN_M=64 python tools/conv_bench.py --iters 20 --only "h64 k64" 2>/dev/null | grep "^c"
P3D_FX_MIN_M=64 python tools/conv_bench.py --iters 20 --only "c64 h64 k256" 2>/dev/null | grep "^c"
