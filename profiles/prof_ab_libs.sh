#!/bin/bash
# kernel-trace stats of profiles/trace_host.py under the old and the current build of the library
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_ab; rm -rf $OUT; mkdir -p $OUT
export KVM_LIB=$PWD/profiles/libkvmatch_hip_old.so
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/old -- python3 profiles/trace_host.py > $OUT/old.log 2>&1
unset KVM_LIB
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/new -- python3 profiles/trace_host.py > $OUT/new.log 2>&1
for v in old new; do
  f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1); echo "== $v"; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s avg %10.1f ns total %12s" % (r['Name'][:70], r['Calls'], float(r['AverageNs']), r['TotalDurationNs']))
PY
done
