R=$PWD; O=$R/gpurun_out/r3m; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "wgrad" > $O/pytest_sel.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
{
python3 tools/ab_flag.py WGRAD_TAP_COLOCATE seg
python3 tools/ab_flag.py WGRAD_TAP_COLOCATE flavr
python3 tools/ab_flag.py WGRAD_TAP_COLOCATE cfg4
} > $O/ab_coloc.txt 2>$O/ab.err; echo "ab rc $?" >> $O/rc.txt
cat $O/rc.txt; cat $O/ab_coloc.txt; tail -2 $O/pytest_sel.log
