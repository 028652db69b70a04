# Round 3: the new workload-level tests (mixed-precision steps, full-size joint steps; REHR_PARITY_FP64=1 also records the
# cfg-4 conditioning yardstick) and the bench with the parity forward moved behind the timed region.
R=$PWD; O=$R/gpurun_out/r3b; mkdir -p $O
python3 -m pytest tests/test_mixed_steps_gpu.py -x -q -m gpu -s > $O/pytest_mixed.log 2>&1; echo "mixed rc $?" >> $O/rc.txt
REHR_PARITY_FP64=${FP64:-1} timeout -k 10 1000 python3 -m pytest tests/test_joint_step_full_size_gpu.py -x -q -m gpu -s > $O/pytest_joint.log 2>&1; echo "joint rc $?" >> $O/rc.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/b20.log 2>$O/b20.err; echo "bench rc $?" >> $O/rc.txt
cat $O/rc.txt
