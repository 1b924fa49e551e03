# timing ablations of the three-piece Winograd weight gradient (DN_WINO_WG_DBG = 1000 + bits; wrong results by construction)
for d in ${DBGS:-0 1001 1002 1004 1008 1016 1024 1032 1015}; do
  echo "== DN_WINO_WG_DBG=$d"
  DN_WGRAD_STREAM=0 DN_WINO_WG_DBG=$d python tools/conv_microbench.py --layers c512_512_16x52,c128_128_64x208 --what wgrad --affine 2>&1 | grep -v amdgpu.ids | tail -2
done
