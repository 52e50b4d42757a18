for v in old p2s p2n p3s; do
  export P3D_LIB=$PWD/variants/lib_$v.so
  for s in "c512 h16 k512 3x3 s1 d1" "c1024 h16 k2048" "c256 h16 k256 3x3" "c256 h16 k1024" "c128 h32 k128 3x3" "c2048 h16 k272"; do
    python tools/conv_bench.py --only "$s" --iters 20 --mode wgrad 2>/dev/null | grep "^c" | sed "s/^/$v  /"
  done
done
export P3D_LIB=$PWD/variants/lib_p2s.so
python -m pytest tests/test_kernels_gpu.py -x -q -k "x3_kernels" > gpurun_out/t_fx.log 2>&1; tail -3 gpurun_out/t_fx.log
