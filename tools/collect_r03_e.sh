R=$PWD; O=$R/gpurun_out/r3f; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for spec in "cfg5 bf16" "seg fp32" "flavr fp32"; do
set -- $spec
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$1 -o k -- python3 $R/bench.py --workload $1 --precision $2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/k_$1.log 2>&1; echo "prof $1 rc $?" >> $O/rc.txt
done
cd $R
B="--no-cpu-baseline --no-kernel-timing --steps 30"
for spec in "seg fp32" "cfg4 fp32" "seg bf16" "flavr fp32"; do
  set -- $spec
  python3 bench.py --workload $1 --precision $2 $B > $O/b_$1_$2.log 2>&1; echo "$1 $2 rc $?" >> $O/rc.txt
done
python3 bench.py --workload cfg5 $B > $O/b_cfg5.log 2>&1; echo "cfg5 rc $?" >> $O/rc.txt
cat $O/rc.txt
