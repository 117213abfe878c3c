#!/bin/bash
# Runs tools/ubench/bin/proc_overlap in 1, 2, 4, 8 and 16 processes at once (on the GPU machine).
B=tools/ubench/bin/proc_overlap
for n in 1 2 4 8 16; do
  echo "== $n processes"
  t0=$(date +%s.%N)
  for i in $(seq $n); do timeout -k 5 60 $B 50 10 1 > /tmp/po_$i.log & done
  wait
  t1=$(date +%s.%N)
  echo "wall $(echo "$t1 - $t0" | bc)"; head -2 /tmp/po_1.log
done
echo "== one process, 4 streams"; timeout -k 5 60 $B 50 12 4
