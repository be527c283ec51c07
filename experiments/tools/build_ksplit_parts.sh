#!/bin/bash
# Builds one diagnostic library per timing ablation of the K-split FILTER kernel (tile 18): experiments/_build/libvodhip_ks<bits>.so =
# the experiment build with kernels_mips_ksplit.hip recompiled with -DKS_ABL=<bits> (1 no epilogue, 2 no hand-off, 4 no LDS-DMA in the loop,
# 8 no step barriers, 16 fragments read once).  Run here (hipcc cross-compiles), the libraries travel to the GPU box with the snapshot.
set -eu
cd "$(dirname "$0")/../.."
make -C vod_amd/csrc ABLATION=1 EXPERIMENTS=1 -j8 > /dev/null
mkdir -p experiments/_build
B=vod_amd/csrc/_build_ablation
OBJS=$(ls $B/*.o | grep -v kernels_mips_ksplit.o)
# an argument is <bits> or <bits>_nf<groups before the middle barrier> (e.g. 0_nf5)
for tag in "$@"; do
  bits=${tag%%_*}; nf=4; [[ $tag == *_nf* ]] && nf=${tag##*_nf}
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DVODHIP_ABLATION -DVODHIP_EXPERIMENTS -DKS_ABL=$bits -DKS_NF=$nf \
    -Ivod_amd/csrc -Iinclude -c experiments/csrc/kernels_mips_ksplit.hip -o experiments/_build/ks_$tag.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs Spill|ScratchSize" | sort | uniq -c
  hipcc --offload-arch=gfx950 -shared -fPIC -o experiments/_build/libvodhip_ks$tag.so $OBJS experiments/_build/ks_$tag.o
  echo built experiments/_build/libvodhip_ks$tag.so
done
