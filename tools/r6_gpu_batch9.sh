#!/bin/bash
# round 6, GPU call 9: final check of the tree as committed -- the whole GPU suite (R3 fixture now present), the default bench line, smoke(),
# the two-rank rehearsal of the N-rank line with the census on plain tensor collectives
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -rP -p no:cacheprovider > gpurun_out/r6_b9_tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r6_b9_tests.txt; grep "R3_5000_golden" gpurun_out/r6_b9_tests.txt | cut -c1-420
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r6_bench_final.json 2> gpurun_out/r6_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_bench_final.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["metric"], d["value"], d["ms_per_step"], "drop-in", d["value_drop_in"], "roofline", r["achieved"], r["frac"], "traffic", r["traffic"], r.get("rocprof_traffic_frac"), "dram", r.get("traffic_dram"),
      "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], {k: v["value"] for k, v in d["secondary"].items()})
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
OCTANE_BENCH_BACKEND=gloo OCTANE_BENCH_ONE_DEVICE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-transfers --no-secondary > gpurun_out/r6_b9_bench2.json 2> gpurun_out/r6_b9_bench2.err
echo "bench2 rc=$?"; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r6_b9_bench2.json') if l.startswith('{')][-1]); print(d['value'], d['n_gpus'], json.dumps(d.get('ranks'))[:500])"
