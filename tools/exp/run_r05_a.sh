#!/bin/bash
# round 5, first box: VALU cost by operand kind; slope of k_y3's time against its vector instruction count
set -u
T="timeout -k 10 150"
echo "##### valu_cost (3 waves per SIMD)"; $T tools/dev/valu_cost 3 2000 600 || exit 1
for rep in 1 2; do
  for ab in 0 16 48 112; do
    echo "##### rep $rep W4_AB=$ab"; $T tools/exp/exp_w4_ab$ab 1048576 64 5 || exit 1
  done
done
echo done
