# timing ablations of the wide three-piece Winograd weight-gradient block (DN_WINO_WG_DBG = 2000 + bits; wrong results by construction)
for d in ${DBGS:-0 2001 2002 2003 2004 2005 2006 2007 2008 2015}; do
  echo "== DN_WINO_WG_DBG=$d"
  DN_WGRAD_STREAM=0 DN_WINO_WG_DBG=$d python tools/conv_microbench.py --layers c512_512_16x52 --what wgrad --affine 2>&1 | grep -v amdgpu.ids | tail -1
done
