cd $GRAFT_REPO_ROOT
for cfg in "GCC_CONCURRENT_TEACHER=0 GCC_OVERLAP_WGRAD=0" "GCC_CONCURRENT_TEACHER=0" "CAPTURE_MODE=thread_local" "CAPTURE_MODE=relaxed"; do
  echo "=== $cfg"; env $cfg timeout 300 python scratch/probe_graph_pix2pix.py 2>&1 | grep -v amdgpu.ids | tail -25
done
