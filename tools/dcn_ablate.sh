#!/bin/bash
# Ablation builds of dcn_bf16.hip (tools only): libsgta_hip_ablK.so with DB_ABLATE=K, timed by tools/dcn_bf16_bench.py
cd "$(dirname "$0")/../sgtapose_amd/csrc"
for K in ${ABLS:-1 2 3 4 5 6 7}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -DDB_ABLATE=$K -c dcn_bf16.hip -o /tmp/dcn_abl$K.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsgta_hip_abl$K.so $(ls *.o | grep -v dcn_bf16.o) /tmp/dcn_abl$K.o
done
