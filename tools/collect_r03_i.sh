R=$PWD; O=$R/gpurun_out/r3j; mkdir -p $O
python3 tools/bench_tconv_ks.py > $O/tconv_intree.txt 2>&1
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "transposed_conv_kernel_equals or phase_interleaved" > $O/pytest_sel.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
{
python3 tools/ab_flag.py USE_TCONV_KS seg
python3 tools/ab_flag.py USE_TCONV_KS cfg5
} > $O/ab_tconv.txt 2>$O/ab.err; echo "ab rc $?" >> $O/rc.txt
grep -v amdgpu $O/tconv_intree.txt; cat $O/rc.txt; cat $O/ab_tconv.txt; tail -3 $O/pytest_sel.log
