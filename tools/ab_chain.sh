#!/bin/bash
# A/B of builds of the library on one box: tools/ab_chain.sh "<other1.so> <other2.so> ..." [chain_bench args]
# (alternating runs: box-to-box and run-to-run spread is 2-3 %, more than most kernel changes are worth)
others=$1; shift
for i in 1 2 3; do
  for o in $others; do
    echo "$(basename $o)  $(timeout -k 10 200 python tools/chain_bench.py --lib "$o" "$@" 2>/dev/null | tail -1 | cut -c85-)"
  done
  echo "tree   $(timeout -k 10 200 python tools/chain_bench.py "$@" 2>/dev/null | tail -1 | cut -c85-)"
done
