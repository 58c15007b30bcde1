mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_layerwise_gpu.py -x -q -m gpu -s 2>&1 | tail -80 > gpurun_out/r04/layerwise_d.txt
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r04/pytest_full_d.txt
tail -5 gpurun_out/r04/layerwise_d.txt; tail -5 gpurun_out/r04/pytest_full_d.txt
