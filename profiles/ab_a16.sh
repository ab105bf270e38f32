#!/bin/bash
# Round 5, class A of the ED pruner on 16 candidates per lane: the A/B runs behind DESIGN 4.2, each on ONE box.
#   ab_a16.sh splits     which window split the launcher picks on full tiles, and the forced ones      -> r05_a16_forced_splits.log
#   ab_a16.sh prefetch   window points fetched 4 (default) / 2 / 1 blocks ahead, and the 8-per-lane form -> r05_a16_prefetch_ab.log
#                        (needs profiles/variants/libkvmatch_hip_ah{2,1}.so: kvm_ed_dot.hip built with -DEDD16_AHEAD=2 / 1,
#                        linked with the other objects of the library; the 8-per-lane form is KVM_ED_NO_A16=1)
#   ab_a16.sh bench      the bench list and full isolated tiles with and without the form, alternating, twice
cd "$(dirname "$0")/.."
one() {  # $1 = KVM_LIB or "", $2 = KVM_ED_NO_A16 or ""
  KVM_ED_NO_A16=$2 KVM_LIB=${1:+$PWD/$1} MS=8192 LS=64,65536 python3 profiles/ed_run_length.py 2>&1 | grep "run length"
  KVM_ED_NO_A16=$2 KVM_LIB=${1:+$PWD/$1} python3 bench.py --steps 5 --warmup 2 --sub-workloads 0 --extras 0 --cpu-baseline 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bench list: value %.4g  kernel_ms %.4f frac %.3f list_bound_ms %.4f full_tiles_ms %.4f' % (d['value'], r['kernel_ms'], r['frac'], r['list_bound_ms'], r['full_tiles_kernel_ms']))"
}
case "${1:-bench}" in
splits)
  for sp in "" 1,1 2,1 4,1; do
    echo "== KVM_ED_SPLIT=${sp:-model}"
    KVM_ED_SPLIT=$sp KVM_TRACE=1 MS=8192 LS=64,65536 python3 profiles/ed_run_length.py 2>&1 | grep -E "run length|ed_dot m" | sort | uniq -c | sort -rn | head -6
  done ;;
prefetch)
  for lib in "" profiles/variants/libkvmatch_hip_ah2.so profiles/variants/libkvmatch_hip_ah1.so; do echo "== lib ${lib:-default (16 per lane, 4 blocks ahead)}"; one "$lib" ""; done
  echo "== 8 per lane (KVM_ED_NO_A16=1)"; one "" 1 ;;
bench)
  for rep in 1 2; do for no in "" 1; do echo "== rep $rep KVM_ED_NO_A16=${no:-unset}"; one "" "$no"; done; done ;;
esac
