#!/bin/bash
# timing-only ablation variants of the bf16 conv kernel (tools/b16_probe.py); results are garbage by construction
set -e
cp unet_amd/libunetmi.so /tmp/libunetmi_orig.so
for n in 0 2 4 8 16 14 30; do
  if [ $n = 0 ]; then cp /tmp/libunetmi_orig.so unet_amd/libunetmi.so; else cp tools/abl/libunetmi_$n.so unet_amd/libunetmi.so; fi
  echo "== ablate $n"
  timeout -k 10 100 python -u tools/b16_probe.py enc4 dec6 dec7 dec8 dec9 2>&1 | grep -v amdgpu.ids
done
cp /tmp/libunetmi_orig.so unet_amd/libunetmi.so
