#!/bin/bash
# usage: tools/ab.sh "<ENV_A>" "<ENV_B>" [rounds]  -- interleaved A/B of bench.py in ONE process group on one box
A="$1"; B="$2"; R=${3:-2}
for i in $(seq $R); do
  for v in "$A" "$B"; do
    echo -n "[$v] "
    env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f clips/s %.2f ms' % (d['value'], d['ms_per_step']))"
  done
done
