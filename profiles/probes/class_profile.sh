#!/bin/bash
# Diagnostic build of the C-ABI library with shader-cycle counters per k_report cost class and per phase of
# d_gen_mapping_report (-DDG_PROFILE_CLASSES; accumulators live in LDS, flushed once per wave).  Run it on the GPU
# box through gpurun only: it rebuilds dart_amd/libdartgpu.so IN THE BOX'S COPY of the repo (which is thrown away
# after the call); the product build has none of this code.
#   gpurun -- 'profiles/probes/class_profile.sh [bench args]'   -> table: one line per class, cycles per 64-read chunk
#   phases: 0 setup | 1 candidate+jobs | 2 seed extension | 3 splice check | 4 normal pairs+validity | 5 classify pairs
#           6 wave-wide NW of large pairs | 7 lane NWs of small pairs | 8 assemble | 9 coordinates+CIGAR out | 10 store
set -e
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DDG_PROFILE_CLASSES -o dart_amd/libdartgpu.so dart_amd/csrc/dg_api.hip
python bench.py --no-cpu-baseline --inflight 1 --steps 1 --warmup 1 "$@" > gpurun_out/class_profile.json 2> gpurun_out/class_profile.err
grep -E "class|chain_heavy" gpurun_out/class_profile.err | tail -30 | tee gpurun_out/class_profile.txt
