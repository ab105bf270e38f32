#!/bin/bash
# the bench lines of the final build, default flags (one per workload)
cd "$(dirname "$0")/.."
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench_rsm_ed.json 2> gpurun_out/final_bench_rsm_ed.err; echo "rsm_ed rc=$?"
for W in cnsm_ed cnsm_dtw mixed; do
  python bench.py --steps 10 --warmup 3 --extras 0 --workload $W > gpurun_out/final_bench_$W.json 2> gpurun_out/final_bench_$W.err; echo "$W rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final_bench_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('final_bench_')[1], 'in flight', d['config']['queries_in_flight'], 'value %.4e ms/query %.4f frac %.3f kernel_ms %.4f abandon %.3e parity %s' % (d['value'], d['ms_per_query'], d['roofline']['frac'], d['roofline']['kernel_ms'], d.get('value_early_abandon', 0), d.get('parity_vs_oracle')))
PY
