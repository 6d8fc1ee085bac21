#!/bin/bash
# rocprofv3 PMC passes (ONE counter per pass, kernel-trace only -- never combined with other trace domains) on the grad_value entry point with
# per-call selection (tools/bench_msda_gv.py, GV_SELECT=1), N = 10 frames (= bench.py: 2 clips x 5 frames), encoder shape:
#   ring     -> the column scatter runs (k_scatter_col4)        trained -> the output-tiled kernels run (k_gv_tile, k_gv_coarse)
# writes gpurun_out/r04_msda_pmc.json (copy to profiles/): per offset pattern and LAUNCHED kernel name FETCH_SIZE / WRITE_SIZE (KB) /
# TCC_EA0_ATOMIC_sum and hbm_bytes_per_launch = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (gfx950 tallies a 128-B read request as 64 B:
# MI355X_MICROARCH.md, section HBM; WRITE_SIZE counts a float atomic as a 64-B request), averaged over the LAST 10 dispatches of each
# kernel (the first calls of a mode run while the selection state settles).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GV_SELECT=1 GV_PATHS= ITERS=12 MSDA_FRAMES=10
for mode in ring trained; do
  for spec in "f FETCH_SIZE" "w WRITE_SIZE" "a TCC_EA0_ATOMIC_sum"; do
    set -- $spec
    rm -rf /tmp/pmcgv_${mode}_$1
    GV_MODES=$mode rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmcgv_${mode}_$1 -- python3 tools/bench_msda_gv.py > gpurun_out/pmcgv_${mode}_$1.log 2>&1 || { tail -5 gpurun_out/pmcgv_${mode}_$1.log; exit 1; }
    F=$(find /tmp/pmcgv_${mode}_$1 -name "*counter_collection.csv" | head -1)
    head -1 $F > gpurun_out/pmcgv_${mode}_$1.csv
    grep -E "k_scatter_col4|k_gv_tile|k_gv_coarse" $F >> gpurun_out/pmcgv_${mode}_$1.csv
  done
done
python3 - <<'PY'
import csv, json, collections
def last(path, ctr, keep=10):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        short = "k_scatter_col4" if "k_scatter_col4" in k else "k_gv_tile" if "k_gv_tile" in k else "k_gv_coarse"
        if r["Counter_Name"] == ctr:
            acc[short].append(float(r["Counter_Value"]))
    return {k: sum(v[-keep:]) / len(v[-keep:]) for k, v in acc.items()}
out = {"note": "rocprofv3 --pmc, one counter per pass (tools/pmc_gv.sh) on tools/bench_msda_gv.py GV_SELECT=1 (ocpg_msda_bwd_value_sel_f32), N = 10 frames "
               "(= bench.py: 2 clips x 5 frames), encoder shape; per offset pattern the kernels that RAN (the other family's launches are idle). "
               "FETCH_SIZE / WRITE_SIZE are KB; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies a 128-B read request as 64 B); WRITE_SIZE counts "
               "a float atomic as a 64-B request; hbm_bytes_per_launch = corrected reads + writes; last 10 dispatches of each kernel averaged",
       "n_frames": 10, "algorithmic_bytes": 182784000, "offsets": {}}
run = {"ring": ("k_scatter_col4",), "trained": ("k_gv_tile", "k_gv_coarse")}
for mode, kernels in run.items():
    f, w, a = (last(f"gpurun_out/pmcgv_{mode}_{c}.csv", n) for c, n in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE"), ("a", "TCC_EA0_ATOMIC_sum")))
    out["offsets"][mode] = {k: {"FETCH_SIZE_KB": f[k], "WRITE_SIZE_KB": w[k], "TCC_EA0_ATOMIC": a[k],
                                "hbm_bytes_per_launch": 2 * f[k] * 1024 + w[k] * 1024} for k in kernels}
json.dump(out, open("gpurun_out/r04_msda_pmc.json", "w"), indent=1)
print(json.dumps(out["offsets"], indent=1))
PY
