mkdir -p gpurun_out
for c in neus nerf hdrnerf; do bash tools/prof_config.sh r5 $c 2>&1 | head -32; done
