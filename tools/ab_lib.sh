#!/bin/bash
# A/B of two builds of libmpqr.so on one GPU lease (boxes differ by +-2.5 %): bash tools/ab_lib.sh <base.so> [bench args]
# Runs new, base, new, base on the default bench.  For use on the GPU box only (it swaps the library in the box's scratch copy of the tree).
base=$1; shift
lib=mixedprecisionblockqr_amd/libmpqr.so
cp $lib /tmp/mpqr_new.so
run() {
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin --no-alone "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); b=d['breakdown_ms']; print('$tag step %.2f panel %.2f far tn/nn %.2f/%.2f q %.2f be %.2e' % (d['ms_per_step'], b['ms_panel'], b['ms_far_tn'], b['ms_far_nn'], b['ms_form_q'], d['error']['backward_error']))
"
}
for rep in 1 2; do
  tag=new;  cp /tmp/mpqr_new.so $lib; run "$@"
  tag=base; cp $base $lib;            run "$@"
done
cp /tmp/mpqr_new.so $lib
