#!/bin/bash
# error metrics of a bench config under environment settings: bash tools/ab_err.sh "<bench args>" "VAR=1" "-" ...
args=$1; shift
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  env $e python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dropin --no-alone $args 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-30s %-14s step %.2f' % ('$args', '$cfg', d['ms_per_step']), {k: (float('%.3e' % v) if isinstance(v, float) else v) for k, v in d['error'].items()})
"
done
