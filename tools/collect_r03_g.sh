R=$PWD; O=$R/gpurun_out/r3h; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_bf16_kernels_gpu.py tests/test_segmodel_golden_gpu.py tests/test_segmodel_gpu.py tests/test_bf16_model_gpu.py -q -m gpu -x > $O/pytest_sel.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
{
python3 tools/ab_flag.py USE_TCONV_KS seg
python3 tools/ab_flag.py USE_TCONV_KS cfg5
} > $O/ab_tconv.txt 2>$O/ab.err; echo "ab rc $?" >> $O/rc.txt
python3 tools/layer_times.py seg > $O/layers_seg.txt 2>&1
cat $O/rc.txt; cat $O/ab_tconv.txt; tail -3 $O/pytest_sel.log
