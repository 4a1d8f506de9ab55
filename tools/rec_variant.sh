#!/bin/bash
# tools/rec_variant.sh <name> [-DPT_ROWS_ABL=n ...]: rec_kernels.hip rebuilt with extra flags, linked with the library's other objects
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/scratch
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -x hip -c pdf_table_amd/csrc/rec_kernels.hip -o tools/scratch/rec_kernels_$name.o
objs=$(ls pdf_table_amd/build/*.o | grep -v rec_kernels.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/scratch/lib_$name.so $objs tools/scratch/rec_kernels_$name.o
echo tools/scratch/lib_$name.so
