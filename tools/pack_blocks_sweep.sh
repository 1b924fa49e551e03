cd /tmp; export TMPDIR=/tmp
for pb in 64 256 512 1024; do
  DN_PACK_BLOCKS=$pb rocprofv3 --kernel-trace --stats -d /tmp/pk$pb -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --alt-steps 0 > /dev/null 2>&1
  echo "== DN_PACK_BLOCKS=$pb"; grep -E "pack" /tmp/pk$pb/p_kernel_stats.csv | cut -d, -f1-4
done
