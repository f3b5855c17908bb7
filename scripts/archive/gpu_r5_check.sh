#!/bin/bash
# Round 5: the GPU suite + smoke + the driver's own command once more on the final tree (host-side changes after the evidence session).
out=gpurun_out/r5_check
mkdir -p $out
export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest_gpu.log)" | tee $out/summary.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $out/summary.txt
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python -c "
import json; j=json.load(open('$out/bench_default.json'))
print('bench default:', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'roofline', j['roofline']['frac'], j['roofline']['traffic'], 'cpu', j['cpu_baseline']['value'])" | tee -a $out/summary.txt
