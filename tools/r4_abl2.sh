cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RGCN_HIP_LIB=$GRAFT_REPO_ROOT/torch-rgcn_amd/torch_rgcn/lib/librgcn_hip_abl.so
for A in 0 64 128 192; do
  RGCN_BWD_BLK_PIPE=0 RGCN_BWD_ABL=$A timeout 300 python tools/kbench.py --what bwd --iters 20 > gpurun_out/r4_abl2_$A.log 2>&1
  echo "abl=$A $(grep -h 'bwd_fused atomic' gpurun_out/r4_abl2_$A.log | sed 's/.*relerr/relerr/') $(grep -h 'per wave' gpurun_out/r4_abl2_$A.log)"
done
