#!/bin/bash
# round 6, GPU call 5: the whole GPU suite after the boundary split (form-against-form tests on the diagnostic library, oracle legs of the
# full-size tests prefetched by a worker thread), then the fixture of R3 at 5000^2, then a two-rank gloo rehearsal of the N-rank bench line
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -rP -x -p no:cacheprovider --durations=15 > gpurun_out/r6_b5_tests.txt 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r6_b5_tests.txt; grep "ORACLE-PREFETCH" gpurun_out/r6_b5_tests.txt | cut -c1-260
grep -A17 "slowest" gpurun_out/r6_b5_tests.txt | head -18
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/golden/make_r3_5000_oracle_golden.py gpurun_out/r3_5000_oracle.npz > gpurun_out/r6_b5_golden.txt 2>&1
echo "golden rc=$?"; tail -3 gpurun_out/r6_b5_golden.txt
OCTANE_BENCH_BACKEND=gloo OCTANE_BENCH_ONE_DEVICE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-transfers --no-secondary > gpurun_out/r6_b5_bench2.json 2> gpurun_out/r6_b5_bench2.err
echo "bench2 rc=$?"; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r6_b5_bench2.json') if l.startswith('{')][-1]); print(d['value'], d['n_gpus'], json.dumps(d.get('ranks'))[:900])"
