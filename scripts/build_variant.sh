#!/bin/bash
# Build a VARIANT of the library for A/B runs inside one GPU call: scripts/build_variant.sh <name> [extra hipcc flags]  ->  coma_amd/_ab/<name>.so
# (load it with COMA_HIP_LIB=coma_amd/_ab/<name>.so; the in-tree library is untouched)
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
mkdir -p coma_amd/_ab/obj_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form"
for f in coma_amd/csrc/*.hip; do
  o=coma_amd/_ab/obj_$NAME/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o coma_amd/_ab/$NAME.so coma_amd/_ab/obj_$NAME/*.o
rm -rf coma_amd/_ab/obj_$NAME
ls -la coma_amd/_ab/$NAME.so
