# timing-only sweep: row-tile count x K slabs on the DiT's four GEMMs at M = 1000 (UA2_GEMM_KSPLIT_HACK gives wrong results)
# needs an experiment build: tools/ubench/build_alt.sh kx ua2_gemm.hip -DUA2_GEMM_EXPERIMENTS, then UA2_LIB=tools/ubench/dbg/libua2hip_kx.so
for bmt in 2 4 8; do for ks in 1 2 3 4 6; do
  echo "=== bmt $bmt ksplit $ks"
  UA2_GEMM_BMT=$bmt UA2_GEMM_KSPLIT_HACK=$ks UA2_SHAPES=dit timeout 120 python tools/ubench/gemm_shapes.py 1000 2>&1 | grep "^M=" | sed 's/skinny.*| tiled/tiled/; s/| row-tiled.*//'
done; done
