python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo\|RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
