mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests/test_generic_conv_gpu.py -q -m gpu -s -k resnet 2>&1 | grep -v "^  \|^$" | tail -40 > gpurun_out/r04/pytest_j.txt
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/r04/pytest_full_j.txt
tail -6 gpurun_out/r04/pytest_j.txt; tail -6 gpurun_out/r04/pytest_full_j.txt
