export VODHIP_LIB=$PWD/vod_amd/csrc/libvodhip_ablation.so
for ph in 0 1 2 3; do echo "phase $ph"; VODHIP_MERGE_PHASE=$ph python3 tools/bench_merge.py 2>&1 | grep "n_shards 8"; done
