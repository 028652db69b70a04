# Round-3 evidence: per-kernel time (rocprofv3 --kernel-trace --stats), HBM traffic (FETCH_SIZE / WRITE_SIZE in separate
# passes) and matrix-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES) for the headline and secondary workloads.
# Run from the repo root on the GPU box: bash tools/collect_profiles_r03.sh [part]   (part in: bench stats pmc stream; default all)
R=$PWD; O=$R/gpurun_out/r3p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PART=${1:-all}
B="--no-cpu-baseline"
# per-kernel evidence (stats, PMC) is collected with the second HIP streams off: a kernel that shares the chip with
# another stream's kernel has neither its own duration nor its own counters
S="--no-wgrad-stream --no-teacher-stream"
prof() { timeout -k 10 300 rocprofv3 "$@"; }
if [ "$PART" = all ] || [ "$PART" = bench ]; then   # the bench lines of every workload, un-profiled, on this one box
python3 $R/bench.py > $O/b_flavr_fp32.log 2>&1 || exit 1; tail -1 $O/b_flavr_fp32.log > $O/b_flavr_fp32.json
for spec in "seg fp32" "cfg4 fp32" "flavr_ref fp32" "flavr bf16" "seg bf16" "cfg4 bf16"; do
  set -- $spec
  python3 $R/bench.py --workload $1 --precision $2 --steps 30 $B > $O/b_$1_$2.log 2>&1 || exit 1; tail -1 $O/b_$1_$2.log > $O/b_$1_$2.json
done
python3 $R/bench.py --workload cfg5 --steps 20 $B > $O/b_cfg5.log 2>&1 || exit 1; tail -1 $O/b_cfg5.log > $O/b_cfg5.json
python3 $R/tools/bench_thin5.py > $O/thin5_bf16.txt 2>&1
python3 $R/tools/bench_feed.py 2>&1 | grep -v "image shape\|Total subjects\|libdrm" > $O/feed_bench.txt
python3 $R/tools/bench_ref_layers.py > $O/ref_layers.txt 2>&1
for w in flavr_ref flavr seg; do python3 $R/tools/layer_times.py $w > $O/layers_$w.txt 2>&1; done
fi
if [ "$PART" = all ] || [ "$PART" = stats ]; then
for w in flavr seg cfg4 flavr_ref; do
  prof --kernel-trace --stats --output-format csv -d $O/k_$w -o k -- python3 $R/bench.py --workload $w --steps 5 --warmup 2 $B $S --no-kernel-timing > $O/k_$w.log 2>&1 || exit 1
done
prof --kernel-trace --stats --output-format csv -d $O/k_seg_bf16 -o k -- python3 $R/bench.py --workload seg --precision bf16 --steps 5 --warmup 2 $B $S --no-kernel-timing > $O/k_seg_bf16.log 2>&1 || exit 1
prof --kernel-trace --stats --output-format csv -d $O/k_cfg5 -o k -- python3 $R/bench.py --workload cfg5 --steps 5 --warmup 2 $B $S --no-kernel-timing > $O/k_cfg5.log 2>&1 || exit 1
prof --kernel-trace --stats --output-format csv -d $O/k_flavr_overlap -o k -- python3 $R/bench.py --workload flavr --steps 5 --warmup 2 $B --no-kernel-timing > $O/k_flavr_overlap.log 2>&1 || exit 1
prof --kernel-trace --stats --output-format csv -d $O/k_flavr_bf16 -o k -- python3 $R/bench.py --workload flavr --precision bf16 --steps 5 --warmup 2 $B $S --no-kernel-timing > $O/k_flavr_bf16.log 2>&1 || exit 1
fi
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
for w in flavr seg cfg5; do
  prof --kernel-trace --output-format csv -d $O/t_$w -o t -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 $B $S --no-kernel-timing > $O/t_$w.log 2>&1 || exit 1
  prof --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_$w -o f -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 $B $S --no-kernel-timing > $O/f_$w.log 2>&1 || exit 1
  prof --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w_$w -o w -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 $B $S --no-kernel-timing > $O/w_$w.log 2>&1 || exit 1
  prof --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/m_$w -o m -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 $B $S --no-kernel-timing > $O/m_$w.log 2>&1 || exit 1
done
prof --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/m_flavr_ref -o m -- python3 $R/bench.py --workload flavr_ref --steps 2 --warmup 1 $B $S --no-kernel-timing > $O/m_flavr_ref.log 2>&1 || exit 1
prof --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/m_seg_bf16 -o m -- python3 $R/bench.py --workload seg --precision bf16 --steps 2 --warmup 1 $B $S --no-kernel-timing > $O/m_seg_bf16.log 2>&1 || exit 1
fi
if [ "$PART" = all ] || [ "$PART" = stream ]; then
prof --kernel-trace --output-format csv -d $O/t_stream -o t -- python3 $R/tools/bench_stream.py --json $O/stream_algo.json > $O/t_stream.log 2>&1 || exit 1
prof --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_stream -o f -- python3 $R/tools/bench_stream.py > $O/f_stream.log 2>&1 || exit 1
prof --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w_stream -o w -- python3 $R/tools/bench_stream.py > $O/w_stream.log 2>&1 || exit 1
fi
cd $R
ls $O
