R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_dyn3; mkdir -p $OUT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 DMVAE_FORCE_DIST=1 DMVAE_PP_DYNAMIC=1
for i in 1 2 3 4 5 6 7 8; do
  MASTER_PORT=$((29600+i)) timeout 300 python bench.py --gpus 1 --steps 40 --warmup 2 --no-cpu-baseline --time-every $(( (i % 2) * 3 + 1 )) > $OUT/run$i.out 2> $OUT/run$i.err; echo "run $i rc=$? $(tail -1 $OUT/run$i.out | cut -c1-120)"
done
