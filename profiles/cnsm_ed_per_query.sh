cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for q in 1 2 6; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tq$q -- python3 profiles/cnsm_ed_per_query.py $q 1 1 2>&1 | grep "abandon"
f=$(find gpurun_out/tq$q -name "*kernel_stats.csv"); python3 - $f <<P
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage'])>0.8: print('   ',r['Name'][:48].ljust(50), r['Calls'].rjust(5), '%9.1f us'%(float(r['AverageNs'])/1e3))
P
done
