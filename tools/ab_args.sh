#!/bin/bash
# A/B of bench.py arguments on config 4: bash tools/ab_args.sh "" "--outer-block 512" ...
for a in "$@"; do
  for rep in 1 2; do
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin --no-alone $a 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); b=d['breakdown_ms']; print('%-28s step %.2f panel %.2f far tn/nn %.2f/%.2f q %.2f be %.2e' % ('$a', d['ms_per_step'], b['ms_panel'], b['ms_far_tn'], b['ms_far_nn'], b['ms_form_q'], d['error']['backward_error']))
"
  done
done
