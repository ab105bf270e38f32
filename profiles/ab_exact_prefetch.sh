#!/bin/bash
# round 5: the in-place exact evaluation of a survivor (one lane adds 8192 terms in order) with the next block's operands fetched
# while the current block is added up (profiles/variants/libkvmatch_hip_exactpf.so) against the build before: the abandoning
# stream, the abandoning pruner alone (rocprofv3, one query in flight), the headline
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
one() {
  KVM_LIB=${1:+$PWD/$1} python3 bench.py --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('${1:-before}: value %.4g  abandon %.4g kernel_ms %.4f' % (d['value'], d['value_early_abandon'], r['kernel_ms']))"
}
for rep in 1 2 3; do one ""; one profiles/variants/libkvmatch_hip_exactpf.so; done
for lib in "" profiles/variants/libkvmatch_hip_exactpf.so; do
  rm -rf gpurun_out/pf_t; KVM_LIB=${lib:+$PWD/$lib} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf_t -- python3 bench.py --in-flight 1 --steps 2 --warmup 1 --sub-workloads 0 --extras 0 --cpu-baseline 0 > /dev/null 2>&1
  python3 -c "
import csv,glob
f=glob.glob('gpurun_out/pf_t/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'ed_dot' in r['Name']: print('${lib:-before}', r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e3,1))
"
done
