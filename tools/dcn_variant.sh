#!/bin/bash
# tools/dcn_variant.sh <name> [-DPT_DCN_MABL=n ...]: lore_kernels.hip rebuilt with extra flags, linked with the library's other objects
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/scratch
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -x hip -c pdf_table_amd/csrc/lore_kernels.hip -o tools/scratch/lore_kernels_$name.o
objs=$(ls pdf_table_amd/build/*.o | grep -v lore_kernels.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/scratch/lib_$name.so $objs tools/scratch/lore_kernels_$name.o
echo tools/scratch/lib_$name.so
