R=$PWD; O=$R/gpurun_out/r3k; mkdir -p $O
python3 tools/bench_tile_order.py > $O/tile_order.txt 2>&1
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "winograd or wino" > $O/pytest_sel.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
{
python3 tools/ab_flag.py WINO_BAND_MAJOR seg
python3 tools/ab_flag.py WINO_BAND_MAJOR flavr
} > $O/ab_band.txt 2>$O/ab.err; echo "ab rc $?" >> $O/rc.txt
grep -v amdgpu $O/tile_order.txt; cat $O/rc.txt; cat $O/ab_band.txt; tail -2 $O/pytest_sel.log
