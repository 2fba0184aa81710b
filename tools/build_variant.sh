# Build variants/libbogp_<tag>.so = the product library with the named sources recompiled under extra flags (A/B and profiling builds;
# the product objects are not touched).  usage: bash tools/build_variant.sh <tag> "<flags>" <source.hip> [...]
# On the GPU box a harness copies the variant over bayesian-optimization_amd/libbogp.so of the scratch tree (tools/probes/run_variants.sh).
set -e
TAG=$1; FLAGS=$2; shift 2
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/bayesian-optimization_amd/csrc
make -s -C $C
mkdir -p $ROOT/variants/obj_$TAG
OBJS=""
for s in $(cd $C && ls *.hip); do
  o=$C/${s%.hip}.o
  for v in "$@"; do
    if [ "$v" = "$s" ]; then
      o=$ROOT/variants/obj_$TAG/${s%.hip}.o
      /opt/rocm/bin/hipcc $FLAGS -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -I/opt/rocm/include -c $C/$s -o $o
    fi
  done
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $OBJS -shared -L/opt/rocm/lib -ldl -Wl,-rpath,/opt/rocm/lib -o $ROOT/variants/libbogp_$TAG.so
ls -la $ROOT/variants/libbogp_$TAG.so
