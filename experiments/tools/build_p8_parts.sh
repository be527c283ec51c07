#!/bin/bash
# Diagnostic libraries of the production 8-phase FILTER kernel with one part removed at compile time (results wrong, timing only):
# experiments/_build/libvodhip_p8_<tag>.so = the ablation build with kernels_mips_8phase.hip recompiled with the given -D flags.
# usage: experiments/tools/build_p8_parts.sh tag1:-DFLAG1 tag2:"-DFLAG2 -DFLAG3" ...
set -eu
cd "$(dirname "$0")/../.."
make -C vod_amd/csrc ABLATION=1 EXPERIMENTS=1 -j8 > /dev/null
mkdir -p experiments/_build
B=vod_amd/csrc/_build_ablation
OBJS=$(ls $B/*.o | grep -v kernels_mips_8phase.o)
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DVODHIP_ABLATION -DVODHIP_EXPERIMENTS $defs \
    -Ivod_amd/csrc -Iinclude -c vod_amd/csrc/kernels_mips_8phase.hip -o experiments/_build/p8_$tag.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs Spill|ScratchSize" | sort | uniq -c
  hipcc --offload-arch=gfx950 -shared -fPIC -o experiments/_build/libvodhip_p8_$tag.so $OBJS experiments/_build/p8_$tag.o
  echo built experiments/_build/libvodhip_p8_$tag.so
done
