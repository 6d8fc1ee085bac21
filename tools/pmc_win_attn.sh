#!/bin/bash
# rocprofv3 PMC passes (one counter per pass, kernel-trace only) on the matrix-core window-attention kernels at the Swin-T stage-1
# shape (644 windows x 245 tokens x 3 heads, bf16): matrix-core busy cycles against vector-ALU active cycles and wave cycles -- the
# numbers behind "softmax-bound" (DESIGN.md section 4.5).  Writes gpurun_out/win_attn_pmc.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WIN_ATTN_MODES=1 WIN_ATTN_ONLY="swin-t stage1"
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU; do
  rm -rf /tmp/pmc_wa_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_wa_$c -- python3 tools/bench_win_attn.py > gpurun_out/pmc_wa_$c.log 2>&1 || { echo "pass $c failed"; tail -3 gpurun_out/pmc_wa_$c.log; continue; }
  F=$(find /tmp/pmc_wa_$c -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && cp $F gpurun_out/pmc_wa_$c.csv
  K=$(find /tmp/pmc_wa_$c -name "*kernel_trace.csv" | head -1)
  [ -n "$K" ] && cp $K gpurun_out/pmc_wa_trace_$c.csv
done
python3 - <<'PY'
import csv, glob, json, collections, os
out = {"note": "rocprofv3 --pmc, one counter per pass (tools/pmc_win_attn.sh) on tools/bench_win_attn.py, Swin-T stage 1 (644 windows x 245 tokens x 3 heads x "
               "head_dim 32, bf16), matrix-core kernels; averages per launch, summed over the chip by rocprofv3.  SQ_VALU_MFMA_BUSY_CYCLES counts "
               "cycles, SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES quad-cycles (MI355X_MICROARCH.md, cycle constants): *_cycles fields are x 4"}
acc = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_wa_SQ*.csv") + glob.glob("gpurun_out/pmc_wa_GRBM*.csv"):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for short in ("k_fwd", "k_bwd_q", "k_bwd_kv"):
            if short + "<" in k:
                per[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (short, c), v in per.items():
        acc[short][c] = sum(v) / len(v)
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_wa_trace_SQ_WAVE_CYCLES.csv"):
    for r in csv.DictReader(open(f)):
        for short in ("k_fwd", "k_bwd_q", "k_bwd_kv"):
            if short + "<" in r["Kernel_Name"]:
                dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    row = dict(d)
    if "SQ_ACTIVE_INST_VALU" in d:
        row["valu_active_cycles"] = 4 * d["SQ_ACTIVE_INST_VALU"]
    if "SQ_WAVE_CYCLES" in d:
        row["wave_cycles"] = 4 * d["SQ_WAVE_CYCLES"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_ACTIVE_INST_VALU" in d:
        row["mfma_busy_over_valu_active"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * d["SQ_ACTIVE_INST_VALU"])
    if dur[k]:
        row["us_per_launch_under_pmc"] = sum(dur[k]) / len(dur[k])
    out[k] = row
json.dump(out, open("gpurun_out/win_attn_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
