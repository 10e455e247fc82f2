#!/bin/bash
# Round-4 headline experiments on ONE box: run on the GPU box,
#   tools/exp/run_r04_headline.sh > gpurun_out/r04_headline.log 2>&1
# (a) two launches with last-arriver finishes vs the three launches that ship, alternating rounds + stamps
# (b) three vs four workgroups per CU on the auto-spectrum loop (and the cross loop at four, spilling)
# (c) the ablation ladder of the shipped kernel down to VALU only
set -u
T="timeout -k 10 120"
echo "##### (a) last-arriver finishes"; $T tools/exp/exp_la 1048576 64 6 || exit 1
echo "##### (a) stamps"; $T tools/exp/exp_la_t 1048576 64 2 || exit 1
echo "##### (a) again (A/B/A/B across processes)"; $T tools/exp/exp_la 1048576 64 4 || exit 1
for rep in 1 2; do
  echo "##### (b) rep $rep: auto loop, three per CU (ships)"; $T tools/exp/exp_auto3 1048576 64 4 || exit 1
  echo "##### (b) rep $rep: auto loop, three per CU, window by buffer loads"; $T tools/exp/exp_auto3g 1048576 64 4 || exit 1
  echo "##### (b) rep $rep: auto loop, four per CU"; $T tools/exp/exp_auto4 1048576 64 4 || exit 1
done
echo "##### (b) auto loop, four per CU, 768 workgroups only (three per CU resident at four-per-CU register budget)"; NEWCHUNKS=12 $T tools/exp/exp_auto4 1048576 64 3 || exit 1
echo "##### (b) cross loop at four per CU (236 B of scratch)"; $T tools/exp/exp_auto4 1048576 64 3 CROSS || exit 1
echo "##### (b) cross loop at three per CU (ships)"; $T tools/exp/exp_auto3 1048576 64 3 CROSS || exit 1
for rep in 1 2; do
  echo "##### (c) rep $rep: full kernel"; $T tools/exp/exp_w4 1048576 64 4 || exit 1
  echo "##### (c) rep $rep: no loads, no LDS traffic (barriers kept)"; $T tools/exp/exp_w4_valu 1048576 64 4 || exit 1
  echo "##### (c) rep $rep: VALU only"; $T tools/exp/exp_w4_valu_nobar 1048576 64 4 || exit 1
done
echo "##### (c) stamped build: clock and lifetimes"; $T tools/exp/exp_w4_t 1048576 64 2 || exit 1
echo done
