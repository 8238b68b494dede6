#!/bin/bash
# Builds oracle/_ref/libref_loaders.so from the REFERENCE's own sources where they lie (nothing is copied):
# the four host loader classes are CUDA-free C++.  The rest of the reference (Engine,
# SmpcController, Utilities) needs nvcc + cuBLAS + cuSOLVER and is unbuildable in this image.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
ref=/root/reference/src
[ -d "$ref" ] || { echo "no $ref: keeping the prebuilt oracle/_ref"; exit 0; }
mkdir -p "$here/_ref"
g++ -std=c++11 -O1 -w -fPIC -shared -I"$ref" \
    -x c++ "$ref/DwnNetwork.cu" "$ref/ScenarioTree.cu" "$ref/Forecaster.cu" "$ref/SmpcConfiguration.cu" \
    -x c++ "$here/ref_loader_shim.cpp" -o "$here/_ref/libref_loaders.so"
echo "built $here/_ref/libref_loaders.so"
# heap-padding preload for the child process that runs the reference's loaders (their constructor overruns two heap
# blocks by 4 bytes, ScenarioTree.cu:66-75); see oracle/malloc_pad.c
gcc -O1 -fPIC -shared "$here/malloc_pad.c" -o "$here/_ref/libmalloc_pad.so"
echo "built $here/_ref/libmalloc_pad.so"
