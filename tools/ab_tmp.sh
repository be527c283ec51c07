bash tools/bench_tiles.sh "--krot 0" 37 32 33 9
bash tools/bench_tiles.sh "--krot 1" 37 32 33 9
