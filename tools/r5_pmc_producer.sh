# counters of the scatter producer (base library, then the slim8 experiment library): where its cycles go
mkdir -p gpurun_out
cd arcnerf_amd/lib; cp libarcnerf_hip.so keep.so; cd ../..
for v in base slim8; do
  cp arcnerf_amd/lib/alt_$v.so arcnerf_amd/lib/libarcnerf_hip.so
  echo "==== $v"
  bash tools/pmc_kernel.sh "bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-configs --no-psnr" scatter_bin_kernel \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES" \
    "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
done
cp arcnerf_amd/lib/keep.so arcnerf_amd/lib/libarcnerf_hip.so; rm arcnerf_amd/lib/keep.so
