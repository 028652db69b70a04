# Round 3: whole GPU suite on the pruned library + the bench with GC diagnostics (20 and 100 steps)
R=$PWD; O=$R/gpurun_out/r3c; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -s > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/b20.log 2>$O/b20.err; echo "bench20 rc $?" >> $O/rc.txt
python3 bench.py --no-cpu-baseline > $O/b100.log 2>$O/b100.err; echo "bench100 rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -5 $O/pytest_all.log
