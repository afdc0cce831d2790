set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2 > $O/smoke.txt; cat $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
(UA2_SHAPES=dit timeout 150 python tools/ubench/gemm_shapes.py 1000; timeout 300 python tools/ubench/gemm_shapes.py 512 1024 2048 6272) 2>&1 | grep M= > $O/gemm_after.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python $R/tools/rocpd_stats.py $(find $O/prof_bench -name "*.db" | head -1) > $O/bench_kernel_stats.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_st2 -o s -- python $R/tools/ubench/prof_legs.py stage2 > $O/st2.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_st2 -name "*.db" | head -1) > $O/stage2_kernel_stats.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c -- python $R/tools/ubench/prof_legs.py config3 > $O/c3.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_c3 -name "*.db" | head -1) > $O/config3_kernel_stats.txt 2>&1
rm -rf $O/prof_bench $O/prof_st2 $O/prof_c3
head -12 $O/bench_kernel_stats.txt | cut -c1-160
