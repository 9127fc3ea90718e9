#!/bin/bash
# A/B of environment settings on config 5 (fp16): bash tools/ab5.sh "VAR=1" ...   ("-" = defaults)
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  env $e python3 bench.py --config c5 --precision fp16 --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-alone 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); b=d['breakdown_ms']; print('%-40s step %.2f factor %.2f panel %.2f q %.2f be %.2e' % ('$cfg', d['ms_per_step'], b['ms_factor'], b['ms_panel'], b['ms_form_q'], d['error']['backward_error']))
"
done
