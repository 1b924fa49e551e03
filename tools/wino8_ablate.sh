#!/bin/bash
# Ablation timings of dn::wino_conv8_kernel's main loop (in-kernel timestamps; results of the ablated variants are wrong by design).
# bits: 16 no split arithmetic | 32 no weight stream | 64 no staging of the next chunk | 256 no matrix instructions | 512 no LDS fragment reads
for d in 4 20 36 68 260 516 84 116 628 276 324 340 372 884; do
  echo "== DN_WINO_DBG=$d"
  DN_WINO8=1 python tools/wino_timing.py $d 2>&1 | grep "loop" | sed 's/prologue.*loop/loop/; s/epilogue.*//'
done
