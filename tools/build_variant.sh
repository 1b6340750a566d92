#!/bin/bash
# Builds one variant of the engine into variants/<name>.so (git-ignored, travels with gpurun): tools/build_variant.sh <name> [extra hipcc flags]
# The object files go to a scratch directory, so the product build in rust-pathtracer_amd/csrc is left alone.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
SRC=$ROOT/rust-pathtracer_amd/csrc; OBJ=/tmp/pt_variant_$name; mkdir -p $OBJ $ROOT/variants
FLAGS="--offload-arch=gfx950 -std=c++17 -O2 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -w $*"
cd $SRC
pids=()
# (per-file flags as in the Makefile: the light-sample kernels without machine LICM; VARIANT_SHADOW_FLAGS overrides)
SHADOW_FLAGS=${VARIANT_SHADOW_FLAGS--mllvm -disable-machine-licm}
for f in pt_engine.hip pt_output.hip pt_compare.hip pt_scene_host.cpp pt_plan.cpp; do /opt/rocm/bin/hipcc $FLAGS -c $f -o $OBJ/${f%.*}.o & pids+=($!); done
/opt/rocm/bin/hipcc $FLAGS ${VARIANT_EXTEND_FLAGS-} -c pt_kern_extend.hip -o $OBJ/pt_kern_extend.o & pids+=($!)
/opt/rocm/bin/hipcc $FLAGS $SHADOW_FLAGS -c pt_kern_shadow.hip -o $OBJ/pt_kern_shadow.o & pids+=($!)
SHADE_FLAGS=${VARIANT_SHADE_FLAGS--mllvm -disable-machine-licm}
SHADE_NT=${VARIANT_SHADE_NT--DPT_QUEUE_NT=1}   # (the non-lean forms: queue words non-temporal, as in the Makefile)
rm -f $OBJ/pt_kern_shade1.o $OBJ/pt_kern_shade4.o
for nl in 1 4; do
  /opt/rocm/bin/hipcc $FLAGS $SHADE_FLAGS -DPT_SHADE_NL=$nl -DPT_SHADE_PART=0 -c pt_kern_shade.hip -o $OBJ/pt_kern_shade${nl}l.o & pids+=($!)
  /opt/rocm/bin/hipcc $FLAGS $SHADE_FLAGS $SHADE_NT -DPT_SHADE_NL=$nl -DPT_SHADE_PART=1 -c pt_kern_shade.hip -o $OBJ/pt_kern_shade${nl}r.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/variants/$name.so $OBJ/*.o
echo built variants/$name.so
