set -x
for tag in base ringA ringB ringC; do
  if [ $tag = base ]; then unset UA2_LIB; else export UA2_LIB=$PWD/tools/ubench/dbg/libua2hip_$tag.so; fi
  echo "=== $tag"
  UA2_SHAPES=dit timeout 300 python tools/ubench/gemm_shapes.py 1000 2>&1 | grep -v "^+" | sed 's/| row-tiled.*//'
  timeout 300 python tools/ubench/gemm_shapes.py 6272 2>&1 | grep -v "^+" | sed 's/| row-tiled.*//'
  timeout 600 python -m pytest tests/test_gpu_invariance.py -x -q 2>&1 | tail -2
done
