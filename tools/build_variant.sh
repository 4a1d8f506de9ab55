#!/bin/bash
# build an experimental copy of the library with extra -D flags: tools/build_variant.sh <name> [-DFOO=1 ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=tools/scratch/lib_$name.so
mkdir -p tools/scratch/obj_$name
objs=""
for f in conv_igemm det_kernels db_model rec_kernels crnn_model lore_kernels lore_model lore_decode lore_processor layout_kernels layout_model dbnas_model cls_kernels graph_ops cvit_model mtl_model mtl_decoder c_api; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -x hip -c pdf_table_amd/csrc/$f.hip -o tools/scratch/obj_$name/$f.o &
  objs="$objs tools/scratch/obj_$name/$f.o"
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c pdf_table_amd/csrc/db_post.cpp -o tools/scratch/obj_$name/db_post.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs tools/scratch/obj_$name/db_post.o
echo $out
