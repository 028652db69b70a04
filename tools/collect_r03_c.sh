# Round 3: whole GPU suite, then the secondary workloads un-profiled and cfg-5 / cfg-3 kernel stats
R=$PWD; O=$R/gpurun_out/r3d; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -s > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
B="--no-cpu-baseline --no-kernel-timing --steps 30"
for spec in "seg fp32" "cfg4 fp32" "flavr_ref fp32" "seg bf16" "cfg4 bf16" "flavr fp32"; do
  set -- $spec
  python3 bench.py --workload $1 --precision $2 $B > $O/b_$1_$2.log 2>&1; echo "$1 $2 rc $?" >> $O/rc.txt
done
python3 bench.py --workload cfg5 $B > $O/b_cfg5.log 2>&1; echo "cfg5 rc $?" >> $O/rc.txt
cd /tmp; export TMPDIR=/tmp
for w in cfg5 seg; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$w -o k -- python3 $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/k_$w.log 2>&1; echo "prof $w rc $?" >> $O/rc.txt
done
cd $R; cat $O/rc.txt; tail -3 $O/pytest_all.log
