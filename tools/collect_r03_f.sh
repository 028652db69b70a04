R=$PWD; O=$R/gpurun_out/r3g; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_bf16_kernels_gpu.py tests/test_mixed_steps_gpu.py tests/test_bf16_model_gpu.py tests/test_segmodel_golden_gpu.py tests/test_train_steps_gpu.py -q -m gpu -x > $O/pytest_sel.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
B="--no-cpu-baseline --no-kernel-timing --steps 30"
for spec in "seg fp32" "cfg4 fp32" "seg bf16" "flavr fp32" "flavr_ref fp32"; do
  set -- $spec
  python3 bench.py --workload $1 --precision $2 $B > $O/b_$1_$2.log 2>&1; echo "$1 $2 rc $?" >> $O/rc.txt
done
python3 bench.py --workload cfg5 $B > $O/b_cfg5.log 2>&1; echo "cfg5 rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pytest_sel.log
