#!/bin/bash
# one line per configuration: bash tools/bench_configs.sh  (c2, c3, c5 fp8, c5 fp16; c4 is bench.py's default)
fmt='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); b=d["breakdown_ms"]; print(d["config"]["workload"], "|", d["dtype"][:14], "| step %.2f factor %.2f panel %.2f q %.2f be %.2e" % (d["ms_per_step"], b["ms_factor"], b["ms_panel"], b["ms_form_q"], d["error"]["backward_error"]))'
for c in c2 c3; do python3 bench.py --config $c --no-cpu-baseline --no-dropin --no-alone 2>/dev/null | python3 -c "$fmt"; done
for p in fp8 fp16; do python3 bench.py --config c5 --precision $p --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-alone 2>/dev/null | python3 -c "$fmt"; done
