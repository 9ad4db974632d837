#!/bin/bash
# Timing-only ablation variants of the bf16 conv kernel (tools/b16_probe.py); their results are garbage by
# construction, so they are NEVER installed as unet_amd/libunetmi.so: each variant is built into tools/abl/ and the
# probe is pointed at it through UNETMI_LIBRARY (unet_amd/_lib.py), which announces the override on stderr.
# Build (CPU box):   tools/abl/run.sh build      Run (GPU box):   tools/abl/run.sh
set -e
cd "$(dirname "$0")/../.."
VARIANTS="2 4 8 16 14 30"
if [ "$1" = build ]; then
  for n in $VARIANTS; do
    # the Makefile refers to ../../include/unetmi.h and writes ../libunetmi.so: give a scratch copy that layout
    r=$(mktemp -d); mkdir -p "$r/pkg/csrc" "$r/include"; cp include/unetmi.h "$r/include/"
    cp unet_amd/csrc/*.hip unet_amd/csrc/*.hpp unet_amd/csrc/Makefile "$r/pkg/csrc/"
    make -C "$r/pkg/csrc" -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -DB16_ABLATE=$n" >/dev/null
    cp "$r/pkg/libunetmi.so" tools/abl/libunetmi_$n.so
    rm -rf "$r"
    echo "built tools/abl/libunetmi_$n.so"
  done
  exit 0
fi
for n in 0 $VARIANTS; do
  echo "== ablate $n"
  if [ $n = 0 ]; then unset UNETMI_LIBRARY; else
    [ -f tools/abl/libunetmi_$n.so ] || { echo "missing tools/abl/libunetmi_$n.so: run '$0 build' first"; exit 1; }
    export UNETMI_LIBRARY=$PWD/tools/abl/libunetmi_$n.so
  fi
  timeout -k 10 100 python -u tools/b16_probe.py enc4 dec6 dec7 dec8 dec9 2>&1 | grep -v amdgpu.ids
done
