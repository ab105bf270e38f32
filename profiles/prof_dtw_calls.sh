#!/bin/bash
# rocprofv3 kernel stats of whole cNSM-DTW calls: the bench list with many answers (trace_dtw.py: eps 10) and three
# selective queries (trace_dtw2.py).  Outputs under gpurun_out/prof_dtw_calls/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_dtw_calls
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/many -- python3 profiles/trace_dtw.py > $OUT/many.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/selective -- python3 profiles/trace_dtw2.py > $OUT/selective.log 2>&1
for t in many selective; do
  f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1)
  echo "== $t"; grep elapsed_ms $OUT/$t.log | cut -c1-220
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if float(r["Percentage"]) > 0.3:
        print("%-70s calls %4d avg %9.1f us  %5.1f %%" % (r["Name"][:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
