python -m pytest tests/test_gpu_graph.py tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -3
