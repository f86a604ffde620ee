# experiments: stem_conv0_kernel<3> with parts switched off (NAF_CONV0_ABL bits: 1 no epilogue, 2 no row stores, 4 no MFMAs)
set -e
cd $GRAFT_REPO_ROOT
for abl in 0 1 3 4 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -Inaf_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1 -DNAF_CONV0_ABL=$abl -c naf_amd/csrc/stem_conv0.hip -o naf_amd/csrc/build/stem_conv0.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o naf_amd/csrc/libnaf_hip.so naf_amd/csrc/build/*.o
  echo "ABL $abl: $(python tools/stem_layer_bench.py 2>&1 | grep 'conv0 3x3')"
done
