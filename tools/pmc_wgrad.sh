R=$PWD; O=$R/gpurun_out/pmcw; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for pm in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  for cfg in "64 64 128 64 64" "512 512 128 16 16"; do
    n=$(echo $cfg | tr ' ' '_')
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pm -d $O/p${i}_$n -o p --output-format csv -- python3 $R/tools/run_one_wgrad.py $cfg > $O/p${i}_$n.log 2>&1 || exit 1
  done
done
ls $O
