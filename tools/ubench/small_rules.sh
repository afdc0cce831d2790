# A/B of two small launcher choices on the legs they touch: 64-row tiles for long-K launches (UA2_GEMM_RULE2), 6 ring slots on the 32-row tile
for v in base rule2 ring6 both; do
  unset UA2_LIB UA2_GEMM_RULE2
  case $v in rule2) export UA2_GEMM_RULE2=1;; ring6) export UA2_LIB=$PWD/tools/ubench/dbg/libua2hip_r2_6.so;; both) export UA2_GEMM_RULE2=1 UA2_LIB=$PWD/tools/ubench/dbg/libua2hip_r2_6.so;; esac
  echo "== $v"
  timeout 200 python tools/ubench/dit_diag.py 2>&1 | grep -v amdgpu | sed -n 6,6p
  timeout 300 python tools/ubench/prof_legs.py batched_both 2>&1 | grep -o "'B': [0-9]*\|decode_ms_per_frame[^,]*"
done
