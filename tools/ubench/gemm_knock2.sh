# GEMM launch alone (operand pre-packed): where the time outside the main loop goes (UA2_GEMM_DBG 16 no epilogue, 32 no main loop)
export UA2_PREPACKED=1
for n in 0 15 16 31 47 48; do
  if [ $n = 0 ]; then unset UA2_LIB; else export UA2_LIB=$PWD/tools/ubench/dbg/libua2hip_gdbg$n.so; fi
  echo "=== dbg $n"
  UA2_SHAPES=dit timeout 120 python tools/ubench/gemm_shapes.py 1000 2>&1 | grep "^M=" | sed 's/skinny.*| tiled/tiled/; s/| row-tiled.*//'
  timeout 120 python tools/ubench/gemm_shapes.py 6272 2>&1 | grep "^M=" | sed 's/skinny.*| tiled/tiled/; s/| row-tiled.*//'
done
