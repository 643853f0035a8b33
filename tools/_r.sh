cd $GRAFT_REPO_ROOT
AB_SKIP_1WAVE=1 AB_REPEAT=3 tools/ab_shade.sh unpack 2>&1 | tee gpurun_out/ab_r05_r.txt
SSDNERF_HIP_LIB=$GRAFT_REPO_ROOT/.variants/unpack/libssdnerf_hip.so python tools/repro_check.py 10 | tail -2 | cut -c1-120
