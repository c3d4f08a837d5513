#!/bin/bash
# Builds the round-4 issue-model micro-benchmarks (profiles/r04a/README.md).  ubench4.hip is hand-written; ubench6 .. ubench13 are
# generated from it in this order (each generator patches the previous stage's source).
cd "$(dirname "$0")"
set -e
for n in 6 7 8 9 10 11 12 13; do python3 gen_ubench$n.py; done
for n in 4 6 7 8 9 10 11 12 13; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-macro-redefined -o ubench$n ubench$n.hip & done
wait
ls ubench4 ubench6 ubench7 ubench8 ubench9 ubench10 ubench11 ubench12 ubench13
