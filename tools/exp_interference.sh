mkdir -p gpurun_out/r3
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin > gpurun_out/r3/exp_$tag.json 2>&1; python3 - <<PY
import json
for l in open("gpurun_out/r3/exp_$tag.json"):
    if l.startswith("{"):
        d=json.loads(l); b=d["breakdown_ms"]; print("$tag", round(d["ms_per_step"],2), "factor", round(b["ms_factor"],2), "panel", round(b["ms_panel"],2), "formq", round(b["ms_form_q"],2), "far tn/nn", round(b["ms_far_tn"],2), round(b["ms_far_nn"],2), "passes", b["n_passes"], "err", d["error"]["backward_error"], d["error"]["q_error_fro"])
PY
}
run fusedcopy A=1
