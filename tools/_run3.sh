python -m pytest tests/test_gpu_graph.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3
bash tools/_sweep.sh "4 32" DN_X=1 DN_X=2
