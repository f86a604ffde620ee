#!/bin/bash
# Build measurement variants of the library that differ in stem_conv1x1.hip only (-DNAF_C1_ABL=<mask>, see the kernel) into
# tools/bin/libnaf_c1_<mask>.so; time them on the GPU box with  NAF_HIP_KNOBS=1 NAF_HIP_LIB=... python tools/stem_layer_bench.py
set -e
cd "$(dirname "$0")/.."
for m in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -Inaf_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1 $EXTRA -DNAF_C1_ABL=$m -c naf_amd/csrc/stem_conv1x1.hip -o /tmp/c1_$m.o
  objs=$(ls naf_amd/csrc/build/*.o | grep -v stem_conv1x1.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o tools/bin/libnaf_c1_$m.so $objs /tmp/c1_$m.o
  echo built tools/bin/libnaf_c1_$m.so
done
