# L2-patch height (UA2_GEMM_GROUP_M) x row tiles per workgroup at the DiT's shapes, GEMM launch alone
export UA2_PREPACKED=1 UA2_SHAPES=dit
for gm in 1 2 4 8 16 32; do for bmt in 0 2 4; do
  echo "=== group_m $gm bmt $bmt"
  if [ $bmt = 0 ]; then unset UA2_GEMM_BMT; else export UA2_GEMM_BMT=$bmt; fi
  UA2_GEMM_GROUP_M=$gm timeout 100 python tools/ubench/gemm_shapes.py 1000 2>&1 | grep "^M=" | sed 's/skinny.*| tiled/tiled/; s/| row-tiled.*//; s/TFLOP.*//'
done; done
