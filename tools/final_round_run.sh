#!/bin/bash
# End-of-round evidence run on the GPU box (gpurun): full GPU suite, bench line, rocprof kernel tables per leg, PMC passes.
# Writes under gpurun_out/final/; copy what should be judged into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
OUT=$R/gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $OUT/pytest_gpu.log
(timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err)
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_codec -- python $R/tools/ubench/codec_decode.py > $OUT/codec_decode.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -- python $R/tools/ubench/codec_decode.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -- python $R/tools/ubench/codec_decode.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -- python $R/bench.py --no-cpu-baseline --no-legs --steps 2 > /dev/null 2>&1
for leg in batched batched256 config3 stage2; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$leg -- python $R/tools/ubench/prof_legs.py $leg > $OUT/leg_$leg.log 2>&1
done
cd $R
python tools/rocpd_stats.py $(find $OUT/prof_codec -name "*.db" | head -1) > $OUT/codec_kernel_stats.txt 2>/dev/null
python tools/rocpd_stats.py $(find $OUT/prof_bench -name "*.db" | head -1) > $OUT/bench_kernel_stats.txt 2>/dev/null
for leg in batched batched256 config3 stage2; do
  python tools/rocpd_stats.py $(find $OUT/prof_$leg -name "*.db" | head -1) > $OUT/${leg}_kernel_stats.txt 2>/dev/null
  rm -rf $OUT/prof_$leg
done
python tools/ubench/pmc_codec.py $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) > $OUT/pmc_codec.txt 2>&1
rm -rf $OUT/prof_codec $OUT/pmc_fetch $OUT/pmc_write $OUT/prof_bench
tail -3 $OUT/pytest_gpu.log; grep decode: $OUT/codec_decode.log; tail -4 $OUT/pmc_codec.txt
