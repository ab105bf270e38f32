#!/bin/bash
# End-of-round refresh: the rocprofv3 recipe, then one bench line per workload (default flags otherwise).
cd "$(dirname "$0")/.."
bash profiles/prof_r02.sh > gpurun_out/prof_r02.log 2>&1
echo "prof done: $(ls gpurun_out/prof_r02 | wc -l) files"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench_rsm_ed.json 2> gpurun_out/final_bench_rsm_ed.err; echo "rsm_ed rc=$?"
for W in cnsm_ed cnsm_dtw mixed; do
  python bench.py --steps 10 --warmup 3 --extras 0 --workload $W > gpurun_out/final_bench_$W.json 2> gpurun_out/final_bench_$W.err; echo "$W rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final_bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('final_bench_')[1], 'value %.4e %s ms/query %.4f frac %.3f kernel_ms %.4f parity %s' % (d['value'], d['unit'], d['ms_per_query'], d['roofline']['frac'], d['roofline']['kernel_ms'], d.get('parity_vs_oracle')))
    except Exception as e: print(f, 'ERR', e)
PY
