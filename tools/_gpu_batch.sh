set -o pipefail
python -m pytest tests/test_round4_gpu.py -x -q -s -m gpu 2>&1 | tee gpurun_out/t_round4.log | tail -5
python tools/ab_config3.py libamber_hip.so libamber_hip_branchless.so 64 2>&1 | tee gpurun_out/r04_ab_branchless.txt | tail -4
AMBER_AMD_LIB=libamber_hip_branchless.so python -m pytest tests/test_config3_parity_gpu.py -x -q -m gpu 2>&1 | tail -3
