cd $GRAFT_REPO_ROOT
for w in 4 2 4 2; do
KVM_ED_WPB=$w timeout -k 10 200 python bench.py --steps 10 --warmup 2 --extras 0 --cpu-baseline 0 --sub-workloads 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('wpb $w value %.4g ms/q %.4f kernel_ms %.4f frac %.3f full_tiles %.3f list_bound %.4f'%(d['value'],d['ms_per_query'],r['kernel_ms'],r['frac'],r['full_tiles_frac_of_peak'],r['list_bound_ms']))" || exit 1
done
