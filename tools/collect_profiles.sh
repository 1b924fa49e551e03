#!/bin/bash
# Copy the judged summaries of a gpu_profile_round*.sh run from gpurun_out/ (scratch) into profiles/ (tracked).
# usage: bash tools/collect_profiles.sh <tag>
tag=${1:?tag}
one() {   # <run tag> <config> <profile stem>
  local t=$1 cfg=$2 stem=$3
  [ -f gpurun_out/bench_$t.json ] || return
  cp gpurun_out/bench_$t.json profiles/${stem}.json
  grep -v amdgpu.ids gpurun_out/bench_$t.err > profiles/${stem/_bench/}_per_layer.txt
  { echo "# DN_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --config $cfg --steps 5 --warmup 2 --profile-steps 0 --no-cpu-baseline --alt-steps 0 (single-stream so that a kernel's duration is its own; vggbn128 = 14 steps with the default launch tape: 2 eager + 2 warm-up + 1 recorded, 1 replay + 1 eager of the bit-for-bit check, 2 warm-up + 5 timed replays; other configs = 7 eager steps)"; head -60 gpurun_out/prof_$t/${t}_kernel_stats.csv; } > profiles/${stem/_bench/}_kernel_stats.csv
  [ -f gpurun_out/pmc_${t}_traffic.json ] && cp gpurun_out/pmc_${t}_traffic.json profiles/${stem/_bench/}_pmc_traffic.json
}
one $tag vggbn128 ${tag}_bench
one ${tag}_photo128 photo128 ${tag}_photo128_bench
one ${tag}_dorn128 dorn128 ${tag}_dorn128_bench
for c in res50_480 vggbn480 dorn128_bf16; do
  [ -f gpurun_out/bench_${tag}_$c.json ] && cp gpurun_out/bench_${tag}_$c.json profiles/${tag}_bench_$c.json
done
[ -f gpurun_out/strong_${tag}.txt ] && cp gpurun_out/strong_${tag}.txt profiles/${tag}_strong_1gpu.txt
[ -f gpurun_out/bench_default_${tag}.json ] && cp gpurun_out/bench_default_${tag}.json profiles/${tag}_bench_default.json
[ -f gpurun_out/bench_${tag}_res50_480_per_layer.txt ] && grep -v amdgpu.ids gpurun_out/bench_${tag}_res50_480_per_layer.txt > profiles/${tag}_res50_480_per_layer.txt
[ -f gpurun_out/tests_${tag}.log ] && tail -5 gpurun_out/tests_${tag}.log > profiles/${tag}_gpu_tests.txt
for f in b4_timeline b4_trace_summary b4_step_gaps; do [ -f gpurun_out/timeline_${tag}/$f.txt ] && cp gpurun_out/timeline_${tag}/$f.txt profiles/${tag}_$f.txt; done
[ -f gpurun_out/sq_${tag}_wino8.txt ] && cp gpurun_out/sq_${tag}_wino8.txt profiles/${tag}_sq_counters_wino8_raw.txt
[ -f gpurun_out/sq_${tag}_wino64.txt ] && cp gpurun_out/sq_${tag}_wino64.txt profiles/${tag}_sq_counters_wino64_raw.txt
ls -la profiles | grep $tag
