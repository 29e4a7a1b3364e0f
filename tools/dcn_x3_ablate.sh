#!/bin/bash
# Ablation builds of dcn_x3.hip (tools only): libsgta_hip_dxK.so with DX_ABLATE=K (bits: 1 no gather reads, 2 no MFMA, 4 window /
# weight DMA for the first chunk only, 8 no blend / split, 16 no weight-fragment reads), timed by
#   SGTA_LIB=sgtapose_amd/csrc/libsgta_hip_dxK.so python tools/dcn_x3_bench.py
cd "$(dirname "$0")/../sgtapose_amd/csrc"
for K in ${ABLS:-1 2 4 8 16}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops \
      -DDX_ABLATE=$K -c dcn_x3.hip -o /tmp/dcn_x3_abl$K.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsgta_hip_dx$K.so $(ls *.o | grep -v dcn_x3.o) /tmp/dcn_x3_abl$K.o
done
