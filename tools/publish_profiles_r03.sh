# gpurun_out/r3p (scratch, written by tools/collect_profiles_r03.sh on the GPU box) -> profiles/r03_* (tracked).
# Run in the build container from the repo root after the gpurun call has merged its outputs.
O=gpurun_out/r3p; P=profiles
for w in flavr seg cfg4 flavr_ref seg_bf16 cfg5 flavr_bf16 flavr_overlap; do
  [ -f $O/k_$w/k_kernel_stats.csv ] && cp $O/k_$w/k_kernel_stats.csv $P/r03_${w}_kernel_stats.csv
done
for f in $O/b_*.json; do
  n=$(basename $f .json); n=${n#b_}
  cp $f $P/r03_bench_${n}_n1.json
done
[ -f $O/thin5_bf16.txt ] && grep -v libdrm $O/thin5_bf16.txt > $P/r03_thin5_layers.txt
[ -f $O/feed_bench.txt ] && cp $O/feed_bench.txt $P/r03_feed_bench.txt
for w in flavr seg cfg5; do
  [ -f $O/f_$w/f_counter_collection.csv ] && python3 tools/pmc_hbm.py $O/t_$w/t_kernel_trace.csv $O/f_$w/f_counter_collection.csv $O/w_$w/w_counter_collection.csv $P/r03_pmc_hbm_$w.json
done
for w in flavr seg cfg5 seg_bf16 flavr_ref; do
  [ -f $O/m_$w/m_counter_collection.csv ] && python3 tools/pmc_mfma.py $O/m_$w/m_counter_collection.csv $O/m_$w/m_kernel_trace.csv $P/r03_pmc_mfma_$w.json
done
if [ -f $O/f_stream/f_counter_collection.csv ]; then
  cp $O/stream_algo.json $P/r03_stream_algorithmic.json
  python3 tools/pmc_hbm.py $O/t_stream/t_kernel_trace.csv $O/f_stream/f_counter_collection.csv $O/w_stream/w_counter_collection.csv $P/r03_pmc_hbm_stream.json --algo $O/stream_algo.json
fi
for w in flavr_ref flavr seg; do [ -f $O/layers_$w.txt ] && grep -v 'amdgpu\|Warning' $O/layers_$w.txt > $P/r03_layer_times_$w.txt; done
[ -f $O/ref_layers.txt ] && grep -v amdgpu $O/ref_layers.txt > $P/r03_ref_shape_layers.txt
for c in 2 3 4 5; do [ -f gpurun_out/parity_cfg$c.json ] && cp gpurun_out/parity_cfg$c.json $P/r03_parity_cfg$c.json; done
ls $P | grep r02
