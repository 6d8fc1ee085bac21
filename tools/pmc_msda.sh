#!/bin/bash
# rocprofv3 PMC passes (one counter per pass, kernel-trace only -- never combined with other trace domains) on the MSDA
# micro-benchmark at N=10 frames (= bench.py: 2 clips x 5 frames, encoder shape, the model's ring-offset sampling pattern);
# writes gpurun_out/pmc_{f,w,a}.csv (counter_collection rows of the MSDA kernels) and gpurun_out/msda_pmc.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MSDA_FRAMES=10 MSDA_MODES=ring MSDA_COLS=1
for spec in "f FETCH_SIZE" "w WRITE_SIZE" "a TCC_EA0_ATOMIC_sum"; do
  set -- $spec
  rm -rf /tmp/pmc_$1
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_$1 -- python3 tools/bench_msda.py > gpurun_out/pmc_$1.log 2>&1 || { tail -5 gpurun_out/pmc_$1.log; exit 1; }
  F=$(find /tmp/pmc_$1 -name "*counter_collection.csv" | head -1)
  head -1 $F > gpurun_out/pmc_$1.csv
  grep -E "k_scatter_col|msda_bwd_gather_row|msda_fwd_fast" $F >> gpurun_out/pmc_$1.csv
done
python3 - <<'PY'
import csv, json, collections
def avg(path, ctr):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        short = "k_scatter_col" if "k_scatter_col" in k else "msda_bwd_gather_row" if "gather_row" in k else "msda_fwd_fast"
        if r["Counter_Name"] == ctr:
            acc[short].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
f, w, a = avg("gpurun_out/pmc_f.csv", "FETCH_SIZE"), avg("gpurun_out/pmc_w.csv", "WRITE_SIZE"), avg("gpurun_out/pmc_a.csv", "TCC_EA0_ATOMIC_sum")
out = {"note": "rocprofv3 --pmc, one counter per pass (tools/pmc_msda.sh) on tools/bench_msda.py, MSDA_FRAMES=10 (= bench.py: 2 clips x 5 frames), "
               "encoder shape, the model's ring-offset sampling pattern; FETCH_SIZE / WRITE_SIZE are KB; read bytes = 2 x FETCH_SIZE x 1024 "
               "(gfx950 tallies a 128-B read request as 64 B: MI355X_MICROARCH.md, section HBM); WRITE_SIZE counts a float atomic as a "
               "64-B request; hbm_bytes_per_launch = corrected reads + writes", "n_frames": 10}
for k in f:
    fk, wk = f[k][0], w.get(k, (0.0, 0))[0]
    out[k] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "TCC_EA0_ATOMIC": a.get(k, (0.0, 0))[0], "launches_averaged": f[k][1],
              "hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024}
json.dump(out, open("gpurun_out/msda_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
