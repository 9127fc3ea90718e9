#!/bin/bash
# config 5 (65536 x 8192, fp16 arithmetic) under bench arguments: bash tools/ab_c5.sh "" "--outer-block 512" ...
for a in "$@"; do
  python3 bench.py --config c5 --precision fp16 --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-alone $a 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); b=d['breakdown_ms']; print('%-28s step %.2f factor %.2f panel %.2f far tn/nn %.2f/%.2f q %.2f be %.2e' % ('$a', d['ms_per_step'], b['ms_factor'], b['ms_panel'], b['ms_far_tn'], b['ms_far_nn'], b['ms_form_q'], d['error']['backward_error']))
"
done
