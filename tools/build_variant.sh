#!/bin/bash
# usage: tools/build_variant.sh <tag> [-D...]  — builds pathtracer_amd/libpt_hip_<tag>.so with extra defines (kernel tuning experiments)
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Iinclude -Ipathtracer_amd/csrc "$@" \
  -o pathtracer_amd/libpt_hip_$tag.so pathtracer_amd/csrc/pt_api.hip 2>&1 | grep -v "warning\|^$" | head -5
