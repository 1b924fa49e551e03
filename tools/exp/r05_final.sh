#!/bin/bash
mkdir -p gpurun_out
exec < /dev/null
python -m pytest tests -q -m gpu > gpurun_out/tests_r05_c.log 2>&1; grep -E "passed|failed|error" gpurun_out/tests_r05_c.log | tail -3
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > gpurun_out/bench_default_r05_c.json 2> gpurun_out/bench_default_r05_c.err
python -c 'import json; l=[json.loads(x) for x in open("gpurun_out/bench_default_r05_c.json").read().splitlines() if x.startswith("{")][-1]; print(l["value"], l["ms_per_step"], l["roofline"]["frac"], l["cpu_baseline"]["value"], l["config"]["launch"])'
