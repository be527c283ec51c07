bash tools/bench_tiles.sh "--nq 128 --k 100" 1 46 47 48 1
bash tools/bench_tiles.sh "--nq 96 --k 100" 1 46 47
bash tools/bench_tiles.sh "--nq 200 --k 100" 0 46 47
