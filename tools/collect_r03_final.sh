# Round 3 final evidence: full GPU suite, the driver's bench command, the torchrun form with one rank (RCCL path),
# then tools/collect_profiles_r03.sh (bench lines of every workload, kernel stats, PMC HBM / MFMA).
R=$PWD; O=$R/gpurun_out/r3z; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.log 2>$O/driver_cmd.err; echo "driver cmd rc $?" >> $O/rc.txt
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/torchrun_n1.log 2>$O/torchrun_n1.err; echo "torchrun n1 rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -2 $O/pytest_all.log
