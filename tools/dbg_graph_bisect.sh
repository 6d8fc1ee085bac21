for cfg in "FULL=1 SYNC=0 KEEP=1"; do
  echo "=== $cfg"
  env $cfg timeout -k 10 200 python -X faulthandler tools/dbg_graph_step.py 8 > "gpurun_out/dbg_gs_tmp_$cfg.txt" 2>&1
  grep -E "^replay|^   first|^captur|^done|Fatal|File \"/root/repo|Error" "gpurun_out/dbg_gs_tmp_$cfg.txt" | cut -c1-300
done
timeout -k 10 300 python bench.py --graph --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_graph.txt 2>&1; grep -n "capture failed\|Error" gpurun_out/bench_graph.txt | head -5 | cut -c1-400; tail -1 gpurun_out/bench_graph.txt | cut -c1-900
