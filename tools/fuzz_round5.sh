cd $GRAFT_REPO_ROOT
timeout -k 10 420 python tools/fuzz_engines.py 1600 7000000 --oracle --lt > gpurun_out/fuzz_r05_1.txt 2>&1; tail -1 gpurun_out/fuzz_r05_1.txt
timeout -k 10 420 python tools/fuzz_engines.py 1600 7100000 --extreme --scaled --oracle > gpurun_out/fuzz_r05_2.txt 2>&1; tail -1 gpurun_out/fuzz_r05_2.txt
timeout -k 10 300 python tools/fuzz_engines.py 1000 7200000 --scaled --lt --stripes > gpurun_out/fuzz_r05_3.txt 2>&1; tail -1 gpurun_out/fuzz_r05_3.txt
timeout -k 10 300 python tools/fuzz_big.py 4 80000 > gpurun_out/fuzz_r05_big1.txt 2>&1; tail -1 gpurun_out/fuzz_r05_big1.txt
timeout -k 10 300 python tools/fuzz_big.py 6 80000 0.03 0.9 > gpurun_out/fuzz_r05_big2.txt 2>&1; tail -1 gpurun_out/fuzz_r05_big2.txt
