#!/bin/bash
# Builds the C/OpenMP CPU restatement (test infrastructure + bench.py's cpu_baseline leg; never the product).
# x86-64-v3 (AVX2 + FMA) instead of -march=native: the library is built in the CPU-only container and travels to the
# GPU box, whose host CPU may differ.
set -e
cd "$(dirname "$0")"
mkdir -p _build
gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC -o _build/libhpvc.so cpu_closed_form.c -lm
echo "built $(pwd)/_build/libhpvc.so"
