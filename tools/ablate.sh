#!/bin/bash
# Timing-only ablations of the x3 K loop (results are WRONG in these builds): builds libp3d_abl_<name>.so next to the product library from p3d_fx.hip with the
# given macros and the product's other objects.  usage: tools/ablate.sh build   (in the build container)   |   tools/ablate.sh run "<conv_bench pattern>"   (GPU box)
cd "$(dirname "$0")/../3d-pose-estimation-with-previleged-information_amd/csrc" || exit 1
declare -A V=( [noread]="-DP3D_TIMING_ONLY_BUILD -DP3D_FX_ABL_NOREAD" [nostage]="-DP3D_TIMING_ONLY_BUILD -DP3D_FX_ABL_NOSTAGE" [nobar]="-DP3D_TIMING_ONLY_BUILD -DP3D_FX_ABL_NOBAR" [noload]="-DP3D_TIMING_ONLY_BUILD -DP3D_FX_ABL_NOLOAD"
               [mfma_only]="-DP3D_TIMING_ONLY_BUILD -DP3D_FX_ABL_NOREAD -DP3D_FX_ABL_NOSTAGE -DP3D_FX_ABL_NOBAR -DP3D_FX_ABL_NOLOAD" [read_mfma]="-DP3D_TIMING_ONLY_BUILD -DP3D_FX_ABL_NOSTAGE -DP3D_FX_ABL_NOBAR -DP3D_FX_ABL_NOLOAD" )
if [ "$1" = build ]; then
  for n in "${!V[@]}"; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -fno-vectorize ${V[$n]} -c p3d_fx.hip -o /tmp/p3d_fx_$n.o &&
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v p3d_fx.o) /tmp/p3d_fx_$n.o -o libp3d_abl_$n.so && echo built $n ) &
  done; wait
else
  cd ../.. ; pat="$2"
  echo "product    $(python tools/conv_bench.py --img --only "$pat" --mode fwd 2>&1 | grep image-fed | head -1)"
  for n in noload nostage nobar noread read_mfma mfma_only; do
    echo "$n   $(P3D_LIB=$PWD/3d-pose-estimation-with-previleged-information_amd/csrc/libp3d_abl_$n.so python tools/conv_bench.py --img --only "$pat" --mode fwd 2>&1 | grep image-fed | head -1)"
  done
fi
