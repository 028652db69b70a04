# (run at the start of round 3, BEFORE the superseded kernel organisations were deleted: the hip_backend switches it
# flips no longer exist; kept as the record of how profiles/r03_ab_superseded.txt was produced)
# Round 3, first call: (1) the driver's 20-step command against the 100-step default,
# (2) A/B of every kernel organisation that round 2 superseded, on the workload that uses it, before their removal.
# bash tools/collect_r03_ab.sh   (GPU box, repo root)
R=$PWD; O=$R/gpurun_out/r3a; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/b20.log 2>$O/b20.err || exit 1
python3 bench.py --no-cpu-baseline > $O/b100.log 2>$O/b100.err || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/b20_nocpu.log 2>$O/b20_nocpu.err || exit 1
python3 -m pytest tests/test_feed_gpu.py tests/test_parallel_gpu.py tests/test_dropin_gpu.py -x -q -m gpu > $O/pytest_small.log 2>&1 || exit 1
{
python3 tools/ab_flag.py USE_WINO_8WAVE flavr
python3 tools/ab_flag.py USE_WGRAD_8WAVE flavr
python3 tools/ab_flag.py USE_WGRAD_TWO_PER_CU flavr
python3 tools/ab_flag.py USE_W32_PIPELINED seg
python3 tools/ab_flag.py USE_WINO_FLAT8 flavr_ref
python3 tools/ab_flag.py USE_HALO_8WAVE seg --precision bf16
python3 tools/ab_flag.py USE_WGRAD_BRICK_8WAVE seg --precision bf16
} > $O/ab_superseded.txt 2>$O/ab.err || exit 1
tail -3 $O/ab_superseded.txt
