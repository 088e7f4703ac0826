for shape in "--cin 64 --cout 64 --dims 23,8,4" "--cin 32 --cout 32 --dims 23,8,4" "--cin 128 --cout 32 --dims 23,8,4"; do
  for lib in default abl; do
    if [ $lib = default ]; then unset CALODIFF_LIB; else export CALODIFF_LIB=$PWD/calodiffusion_amd/lib/libcalodiff_hip_abl.so; fi
    echo "$lib $shape: $(python tools/conv_bench.py --batch 64 $shape --iters 30 2>/dev/null | grep 'kernel us')"
  done
done
