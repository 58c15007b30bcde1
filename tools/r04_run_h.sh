mkdir -p gpurun_out/r04
python tools/diag_resnet.py f32 > gpurun_out/r04/diag_resnet_f32.txt 2>&1
python tools/diag_resnet.py bf16 > gpurun_out/r04/diag_resnet_bf16.txt 2>&1
timeout 1500 python -m pytest tests/test_generic_conv_gpu.py -q -m gpu -k "conv_fn or pools" 2>&1 | grep -v "^  \|^$" | head -80 > gpurun_out/r04/pytest_h.txt
tail -3 gpurun_out/r04/pytest_h.txt
