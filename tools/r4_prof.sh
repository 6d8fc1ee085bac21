#!/bin/bash
# steady-state per-step kernel summary of the default bench command (rocprofv3 --kernel-trace --stats) + the per-module op census
tag=${1:-a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_prof_$tag.log 2>&1 || { tail -5 gpurun_out/r4/bench_prof_$tag.log; exit 1; }
F=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_trace.py $F gpurun_out/r4/steady_$tag.csv 3 k_scatter_col 4
S=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/r4/kernel_stats_$tag.csv
python3 - gpurun_out/r4/steady_$tag.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(l for l in open(sys.argv[1]) if not l.startswith('"#')))
fam = {}
def family(n):
    if n.startswith("void at::native") or "at::native" in n or "at::cuda" in n or "elementwise_kernel" in n or "reduce_kernel" in n or "CatArray" in n: return "ATen"
    if n.startswith("Cijk_") or "Custom_Cijk" in n: return "hipBLASLt"
    if "igemm" in n or "ck::" in n or "_ZN2ck" in n or "SubTensorOp" in n or "miopen" in n.lower(): return "MIOpen/CK"
    if "fft_rtc" in n or "transpose_rtc" in n or "rocfft" in n.lower(): return "rocFFT"
    if "rocclr" in n: return "runtime copies/fills"
    return "own HIP"
for r in rows:
    f = family(r["Name"]); a = fam.setdefault(f, [0.0, 0.0]); a[0] += float(r["MsPerStep"]); a[1] += float(r["CallsPerStep"])
for f, (ms, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print("%-22s %7.2f ms/step %7.0f launches/step" % (f, ms, n))
print("---- top 40")
for r in rows[:40]:
    print("%-100s %6s x %8s us = %7s ms" % (r["Name"][:100], r["CallsPerStep"], r["AvgUs"], r["MsPerStep"]))
PY
