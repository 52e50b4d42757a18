for v in base m16; do
  if [ $v = base ]; then unset P3D_LIB; else export P3D_LIB=$PWD/variants/lib_$v.so; fi
  python tools/conv_bench.py --iters 30 --mode wgrad 2>/dev/null | grep "^c\|total" | sed "s/^/$v  /" | cut -c1-44,95-130
done
export P3D_LIB=$PWD/variants/lib_m16.so
python -m pytest tests/test_kernels_gpu.py -x -q -k "x3_kernels or sampled" 2>&1 | tail -2
