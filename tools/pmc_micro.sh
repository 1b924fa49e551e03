#!/bin/bash
# SQ / GRBM counters for single conv layers (tools/conv_microbench.py), one counter group per pass.
# usage (GPU box): bash tools/pmc_micro.sh <tag> <layers> <what>
# 4th argument "ta": the texture-addresser / L1 groups only (TA busy, wavefronts, stalls; TCP accesses; TD busy)
tag=${1:-x}; layers=${2:-c128_128_64x208}; what=${3:-fwd}; sel=${4:-sq}
R=$(pwd); out=$R/gpurun_out/pmcm_$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
if [ "$sel" = "ta" ]; then
# (two counters per block and pass: a group the hardware cannot collect aborts rocprofv3, which then hangs in its signal handler)
for grp in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $grp -d $out/pass$i -o p --output-format csv -- python $R/tools/conv_microbench.py --reps 5 --batch ${BATCH:-32} --layers $layers --what $what > $out/pass$i.log 2>&1 || echo "pass $i failed"
done
else
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $grp -d $out/pass$i -o p --output-format csv -- python $R/tools/conv_microbench.py --reps 5 --batch ${BATCH:-32} --layers $layers --what $what > $out/pass$i.log 2>&1 || echo "pass $i failed"
done
fi
cd $R
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int); dur = collections.defaultdict(float)
for f in glob.glob("$out/pass*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "igemm" not in k and "wino" not in k and "lds3" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen and r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"):
            seen.add(key)
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1; dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, v in agg.items():
    print(k)
    n = max(cnt[k], 1)
    for c, x in sorted(v.items()): print("   %-28s %16.0f per launch" % (c, x / n))
    if "GRBM_GUI_ACTIVE" in v and dur[k] > 0: print("   effective clock %.3f GHz, avg duration %.1f us" % (v["GRBM_GUI_ACTIVE"] / dur[k], dur[k] / n / 1e3))
    if "TA_TA_BUSY_sum" in v and "GRBM_GUI_ACTIVE" in v: print("   TA busy = %.3f of (256 CUs x cycles)" % (v["TA_TA_BUSY_sum"] / (256 * v["GRBM_GUI_ACTIVE"] / 8)))
    if "TD_TD_BUSY_sum" in v and "GRBM_GUI_ACTIVE" in v: print("   TD busy = %.3f" % (v["TD_TD_BUSY_sum"] / (256 * v["GRBM_GUI_ACTIVE"] / 8)))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v: print("   MFMA busy / SQ busy = %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CYCLES"]))
PY
