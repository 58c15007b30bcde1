mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_generic_conv_gpu.py -q -m gpu -s 2>&1 | tail -60 > gpurun_out/r04/pytest_g.txt
timeout 1500 python -m pytest tests/test_backbone_gpu.py tests/test_dist_gpu.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r04/pytest_g2.txt
timeout 1500 python -m pytest tests/test_configs_gpu.py -q -m gpu -x -s -k "cfg0" 2>&1 | tail -25 > gpurun_out/r04/pytest_g3.txt
tail -4 gpurun_out/r04/pytest_g.txt; tail -3 gpurun_out/r04/pytest_g2.txt; tail -3 gpurun_out/r04/pytest_g3.txt
