# Round 3: the reworked tests + A/B of the phase interleave and the un-profiled secondary benches on the current library
R=$PWD; O=$R/gpurun_out/r3e; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_mixed_steps_gpu.py tests/test_inference_gpu.py tests/test_bf16_kernels_gpu.py tests/test_kernels_gpu.py tests/test_bf16_model_gpu.py tests/test_joint_step_full_size_gpu.py -q -m gpu -s > $O/pytest_sel.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
{
python3 tools/ab_flag.py PHASE_INTERLEAVE seg
python3 tools/ab_flag.py PHASE_INTERLEAVE cfg5
python3 tools/ab_flag.py PHASE_INTERLEAVE flavr
} > $O/ab_interleave.txt 2>$O/ab.err; echo "ab rc $?" >> $O/rc.txt
B="--no-cpu-baseline --no-kernel-timing --steps 30"
for spec in "seg fp32" "cfg4 fp32" "flavr_ref fp32" "seg bf16" "flavr fp32"; do
  set -- $spec
  python3 bench.py --workload $1 --precision $2 $B > $O/b_$1_$2.log 2>&1; echo "$1 $2 rc $?" >> $O/rc.txt
done
python3 bench.py --workload cfg5 $B > $O/b_cfg5.log 2>&1; echo "cfg5 rc $?" >> $O/rc.txt
cat $O/rc.txt; cat $O/ab_interleave.txt; tail -3 $O/pytest_sel.log
