R=$PWD; O=$R/gpurun_out/r3i; mkdir -p $O
python3 tools/bench_tconv_ks.py > $O/tconv_intree.txt 2>&1
REHRSEG_HIP_LIB=$R/tools/_alt/lib_NOSTORE.so python3 tools/bench_tconv_ks.py > $O/tconv_nostore.txt 2>&1
REHRSEG_HIP_LIB=$R/tools/_alt/lib_NOMFMA.so python3 tools/bench_tconv_ks.py > $O/tconv_nomfma.txt 2>&1
timeout -k 10 300 python3 -m pytest tests/test_bf16_kernels_gpu.py -q -m gpu -x -k "instnorm_backward" > $O/pytest_sel.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
grep -v amdgpu $O/tconv_intree.txt; grep -v amdgpu $O/tconv_nostore.txt; grep -v amdgpu $O/tconv_nomfma.txt; cat $O/rc.txt
