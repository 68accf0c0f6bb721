python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3_t8_tests.txt
python -m pytest tests/test_r3_gpu.py -m gpu -q -s -k "loss or grad" 2>&1 | grep "^\[" > gpurun_out/r3_t8_detail.txt
python - > gpurun_out/r3_t8_spectral.txt 2>&1 <<'P'
import torch, bench
print(bench.spectral_cluster_bench(bench.CLUSTER_SHAPES["cfg2"], torch.device("cuda")))
P
cat gpurun_out/r3_t8_tests.txt gpurun_out/r3_t8_detail.txt gpurun_out/r3_t8_spectral.txt
