#!/bin/bash
# A16: which window split does the launcher pick on full tiles, and what do the forced ones give?
for sp in "" 1,1 2,1 4,1; do
  echo "== KVM_ED_SPLIT=${sp:-model}"
  KVM_ED_SPLIT=$sp KVM_TRACE=1 MS=8192 LS=64,65536 python3 profiles/ed_run_length.py 2>&1 | grep -E "run length|ed_dot m" | sort | uniq -c | sort -rn | head -6
done
