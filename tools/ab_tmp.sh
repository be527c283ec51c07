bash tools/bench_tiles.sh "" 9 10 9 10
bash tools/bench_tiles.sh "--nq 512" 9 10
bash tools/bench_tiles.sh "--nq 256" 9 10
