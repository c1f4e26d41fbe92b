#!/bin/bash
# round 5, GPU call 17: final check -- the default bench line (traffic from the committed counter passes), the whole GPU suite, smoke()
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_bench_final.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["metric"], d["value"], d["ms_per_step"], "drop-in", d["value_drop_in"], "roofline", r["achieved"], r["frac"], "traffic", r["traffic"], r["rocprof_traffic_frac"], "copy", r["frac_of_measured_copy"],
      "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], {k: v["value"] for k, v in d["secondary"].items()})
PY
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r5_b17_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5_b17_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
