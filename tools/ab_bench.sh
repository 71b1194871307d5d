#!/bin/bash
# interleaved A/B of two environments on ONE box: tools/ab_bench.sh "ENV_A=1" "ENV_B=1" [pairs]
a="$1"; b="$2"; n=${3:-2}
for i in $(seq $n); do
  for e in "$a" "$b"; do
    r=$(env $e python bench.py --no-cpu-baseline --no-serve --steps 8 --warmup 2 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'])")
    echo "[$e] $r"
  done
done
