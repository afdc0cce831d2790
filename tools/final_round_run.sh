#!/bin/bash
# End-of-round evidence run on the GPU box (gpurun): full GPU suite, bench line (plain and under torchrun with one rank: the RCCL path),
# rocprof kernel tables per leg, PMC passes (codec chain, the roofline GEMV, the order-free GEMM).
# Writes under gpurun_out/final/; copy what should be judged into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
OUT=$R/gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $OUT/pytest_gpu.log
(timeout 1200 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err)
# one rank under torchrun: init_process_group("nccl"), all_gather and barrier execute on this box (VERDICT r4 #6a)
(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-legs --config4-leg --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_line_torchrun_1rank.json 2> $OUT/bench_torchrun.err)
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_codec -- python $R/tools/ubench/codec_decode.py > $OUT/codec_decode.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -- python $R/tools/ubench/codec_decode.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -- python $R/tools/ubench/codec_decode.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_sw_fetch -- python $R/tools/ubench/pmc_swiglu.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_sw_write -- python $R/tools/ubench/pmc_swiglu.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_g2_a -- python $R/tools/ubench/pmc_gemm2.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $OUT/pmc_g2_b -- python $R/tools/ubench/pmc_gemm2.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_g2_c -- python $R/tools/ubench/pmc_gemm2.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -- python $R/bench.py --no-cpu-baseline --no-legs --steps 2 > /dev/null 2>&1
for leg in batched batched256 config3 stage2; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$leg -- python $R/tools/ubench/prof_legs.py $leg > $OUT/leg_$leg.log 2>&1
done
# round 6: the order-free legs and the one-window DiT step, launch by launch
UA2_ORDER_FREE_ROWS=2048 rocprofv3 --kernel-trace --stats -d $OUT/prof_config3_of -- python $R/tools/ubench/prof_legs.py config3 > $OUT/leg_config3_of.log 2>&1
UA2_ORDER_FREE_ROWS=1024 rocprofv3 --kernel-trace --stats -d $OUT/prof_batched1024_of -- python $R/tools/ubench/prof_legs.py batched1024 > $OUT/leg_batched1024_of.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_dit -- python $R/tools/ubench/dit_step_profile.py 20 > $OUT/leg_dit.log 2>&1
cd $R
python tools/rocpd_stats.py $(find $OUT/prof_codec -name "*.db" | head -1) > $OUT/codec_kernel_stats.txt 2>/dev/null
python tools/rocpd_stats.py $(find $OUT/prof_bench -name "*.db" | head -1) > $OUT/bench_kernel_stats.txt 2>/dev/null
for leg in batched batched256 config3 stage2 config3_of batched1024_of; do
  python tools/rocpd_stats.py $(find $OUT/prof_$leg -name "*.db" | head -1) > $OUT/${leg}_kernel_stats.txt 2>/dev/null
  rm -rf $OUT/prof_$leg
done
python tools/rocpd_stats.py $(find $OUT/prof_dit -name "*.db" | head -1) --by-grid > $OUT/dit_step_kernel_stats.txt 2>/dev/null
rm -rf $OUT/prof_dit
python tools/ubench/pmc_codec.py $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) > $OUT/pmc_codec.txt 2>&1
python tools/ubench/pmc_swiglu_report.py $(find $OUT/pmc_sw_fetch -name "*.db" | head -1) $(find $OUT/pmc_sw_write -name "*.db" | head -1) > $OUT/pmc_swiglu.txt 2>&1
{ echo "# pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; python tools/pmc_parse.py $(find $OUT/pmc_g2_a -name "*.db" | head -1) gemm2_kernel;
  echo "# pass 2: SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; python tools/pmc_parse.py $(find $OUT/pmc_g2_b -name "*.db" | head -1) gemm2_kernel;
  echo "# pass 3: FETCH_SIZE"; python tools/pmc_parse.py $(find $OUT/pmc_g2_c -name "*.db" | head -1) gemm2_kernel; } > $OUT/pmc_gemm2.txt 2>&1
rm -rf $OUT/prof_codec $OUT/pmc_fetch $OUT/pmc_write $OUT/prof_bench $OUT/pmc_sw_fetch $OUT/pmc_sw_write $OUT/pmc_g2_a $OUT/pmc_g2_b $OUT/pmc_g2_c
tail -3 $OUT/pytest_gpu.log; grep decode: $OUT/codec_decode.log; tail -4 $OUT/pmc_codec.txt; tail -3 $OUT/pmc_swiglu.txt; head -6 $OUT/pmc_gemm2.txt
