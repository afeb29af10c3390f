#!/bin/bash
# Dev tool: a second libmusica_hip built with extra compiler flags / defines, for same-box A/B runs of kernel forms.
#   bash devtools/build_variant.sh NAME "<extra hipcc flags>"  ->  <pkg>/libmusica_hip_NAME.so   (PROBE_LIB=<that path> devtools/linear_probe.py ...)
set -e
NAME=$1; EXTRA=$2
PKG=$(dirname $0)/../metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd
OBJ=$PKG/build/var_$NAME
mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-value -Wno-unused-result"
pids=()
for f in kernels_pyramid kernels_expand_sd kernels_analysis kernels_gradation kernels_clahe kernels_bench musica_ctx; do
    NOSLP=""; case $f in kernels_analysis|kernels_expand_sd) NOSLP="-fno-slp-vectorize";; esac   # as build.py's NO_SLP
    /opt/rocm/bin/hipcc $FLAGS $NOSLP $EXTRA -x hip -c $PKG/csrc/$f.hip -o $OBJ/$f.o &
    pids+=($!)
done
/opt/rocm/bin/hipcc $FLAGS -c $PKG/csrc/musica_io.cpp -o $OBJ/musica_io.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libmusica_hip_$NAME.so $OBJ/*.o
echo $PKG/libmusica_hip_$NAME.so
