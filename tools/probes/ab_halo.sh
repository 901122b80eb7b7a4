# Same-box A/B of the kx-halo form of conv_pp (DMVAE_PP_HALO).  usage: bash tools/probes/ab_halo.sh [values...]
R=$GRAFT_REPO_ROOT; cd $R
for v in "$@"; do echo "=== DMVAE_PP_HALO=$v"; DMVAE_PP_HALO=$v REPS=${REPS:-20} python tools/bench_conv.py fwd 2>&1 | grep -v "Warn\|amdgpu.ids"; done
