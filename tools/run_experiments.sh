#!/bin/bash
# Times every kernel variant built by tools/conv_experiments.py (GPU box).
for v in $(python3 tools/conv_experiments.py list); do
  echo "== $v"
  SRHIP_LIBRARY=$PWD/superresolution_amd/_exp/libsrhip_$v.so timeout -k 10 120 python3 -u tools/bench_conv.py one 2>&1 | grep -v amdgpu.ids || exit 1
done
