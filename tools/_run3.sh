python -m pytest tests/test_gpu_graph.py tests/test_gpu_two_ranks.py tests/test_gpu_rccl.py -x -q -m gpu 2>&1 | tail -4
bash tools/_sweep.sh "4 8 32" DN_WGRAD_STREAMS=1 DN_WGRAD_STREAMS=2 DN_WGRAD_STREAMS=3 DN_WGRAD_STREAMS=1 DN_WGRAD_STREAMS=2
