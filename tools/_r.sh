cd $GRAFT_REPO_ROOT
AB_SKIP_1WAVE=1 AB_REPEAT=3 tools/ab_shade.sh we1 we1hg8 2>&1 | tee gpurun_out/ab_r05_q.txt
for v in we1 we1hg8; do SSDNERF_HIP_LIB=$GRAFT_REPO_ROOT/.variants/$v/libssdnerf_hip.so python tools/repro_check.py 8 | tail -2 | cut -c1-100; done
