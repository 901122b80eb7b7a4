R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_dyn; mkdir -p $OUT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 DMVAE_FORCE_DIST=1
DMVAE_PP_DYNAMIC=1 timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --time-every 1 > $OUT/dyn1.out 2> $OUT/dyn1.err; echo "rc=$?"
grep -v "^frame\|^$" $OUT/dyn1.err | head -30
DMVAE_PP_DYNAMIC=1 timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --time-every 4 > $OUT/dyn2.out 2> $OUT/dyn2.err; echo "rc=$? (time-every 4)"
unset RANK WORLD_SIZE LOCAL_RANK DMVAE_FORCE_DIST
DMVAE_PP_DYNAMIC=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --time-every 1 > $OUT/dyn3.out 2> $OUT/dyn3.err; echo "rc=$? (no dist, dyn, time-every 1)"; tail -c 300 $OUT/dyn3.out
