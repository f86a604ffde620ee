#!/bin/bash
# Variants of the 3x3 key-pooling schedule (tools/gen_stem_rows.py knobs given as VAR=value words, e.g. NAF_ROWS_POOL_DIST=6 =
# slots between a chain's LDS reads and its MFMA).  Builds tools/bin/libnaf_pv<i>.so from a private copy of the sources
# (the .inc is found beside the header that includes it).  usage: tools/pool_dist.sh "NAF_ROWS_POOL_DIST=4" "NAF_ROWS_POOL_DIST=6 NAF_ROWS_POOL_EXTRA=8" ...
set -eo pipefail
cd "$(dirname "$0")/.."
mkdir -p /tmp/pdist/src /tmp/pdist/obj tools/bin
cp naf_amd/csrc/build/*.o /tmp/pdist/obj/
cp naf_amd/csrc/*.h naf_amd/csrc/stem_conv.hip /tmp/pdist/src/
i=0
for kv in "$@"; do
  i=$((i+1))
  env $kv python tools/gen_stem_rows.py /tmp/pdist/src/stem_rows_sched.inc | tail -1 | cut -c1-50
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -I/tmp/pdist/src -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -c /tmp/pdist/src/stem_conv.hip -o /tmp/pdist/obj/stem_conv.o 2>&1 | grep "VGPRs Spill" | head -1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o tools/bin/libnaf_pv$i.so /tmp/pdist/obj/*.o
  echo "built tools/bin/libnaf_pv$i.so: $kv"
done
