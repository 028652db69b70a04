R=$PWD; O=$R/gpurun_out/r1k; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/write.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kf -o k -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kf.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o k -- python3 $R/bench.py --workload seg --steps 5 --warmup 2 --no-cpu-baseline > $O/ks.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kr -o k -- python3 $R/bench.py --workload flavr_ref --steps 5 --warmup 2 --no-cpu-baseline > $O/kr.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k4 -o k -- python3 $R/bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline > $O/k4.log 2>&1 &&
cd $R && timeout -k 10 400 python bench.py --steps 10 --warmup 3 > $O/bench_flavr.log 2>&1 &&
timeout -k 10 300 python bench.py --workload seg --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_seg.log 2>&1 &&
timeout -k 10 300 python bench.py --workload flavr_ref --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ref.log 2>&1 &&
timeout -k 10 300 python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.log 2>&1
ls $O
