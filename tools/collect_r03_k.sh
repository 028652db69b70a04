R=$PWD; O=$R/gpurun_out/r3l; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-timing"
prof() { timeout -k 10 300 rocprofv3 "$@"; }
for w in seg flavr; do
  prof --kernel-trace --output-format csv -d $O/t_$w -o t -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 $B > $O/t_$w.log 2>&1 || exit 1
  prof --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_$w -o f -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 $B > $O/f_$w.log 2>&1 || exit 1
  prof --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w_$w -o w -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 $B > $O/w_$w.log 2>&1 || exit 1
done
ls $O
