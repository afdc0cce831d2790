# timing-only knock-outs of the tiled GEMM main loop (UA2_GEMM_DBG bits: 1 no refills, 2 no MFMA, 4 no fragment reads, 8 no barrier)
for n in 0 1 2 4 8 3 6 7 15; do
  if [ $n = 0 ]; then unset UA2_LIB; else export UA2_LIB=$PWD/tools/ubench/dbg/libua2hip_gdbg$n.so; fi
  echo "=== dbg $n"
  UA2_SHAPES=dit timeout 120 python tools/ubench/gemm_shapes.py 1000 2>&1 | grep "^M=" | sed 's/skinny.*| tiled/tiled/; s/| row-tiled.*//'
  timeout 120 python tools/ubench/gemm_shapes.py 6272 2>&1 | grep "^M=" | sed 's/skinny.*| tiled/tiled/; s/| row-tiled.*//'
done
