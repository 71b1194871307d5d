#!/bin/bash
# interleaved comparison of several environments on ONE box: tools/ab3.sh rounds "ENV=a" "ENV=b" ...
n=$1; shift
for i in $(seq $n); do
  for e in "$@"; do
    r=$(env $e python bench.py --no-cpu-baseline --no-serve --steps 8 --warmup 2 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'])")
    echo "[$e] $r"
  done
done
