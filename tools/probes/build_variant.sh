# Library variant for same-box A/Bs: rebuild ONE source with extra defines and link it with the in-tree objects.  usage: bash tools/probes/build_variant.sh <name> <source.hip> "<-Dflags>"
R=$(cd $(dirname $0)/../.. && pwd); N=$1; SRC=$2; FLAGS=$3
mkdir -p $R/tools/probes/bin
cd $R/dmvae_amd/csrc && make -s >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-value -Wno-c++20-extensions $FLAGS -c $SRC -o /tmp/variant_$N.o || exit 1
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/variant_$N.o -o $R/tools/probes/bin/lib_$N.so && echo built lib_$N.so
