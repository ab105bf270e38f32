#!/usr/bin/env python3
"""Lint of gfx950 assembly for a miscompile met in round 5 (kvm_ed_dot.hip, hipcc of ROCm 7.2): a wave-uniform condition
held in a VGPR is compared with v_cmp (result in VCC) and then consumed by s_cselect_b32, which reads SCC -- the
`s_and_b64 vcc, vcc, exec` that should set SCC between the two was dropped, so the select took whatever SCC the exec-mask
restore in front of it had left (always 1).  Flags every SCC reader (s_cselect_*, s_cbranch_scc*, s_cmov*) whose nearest
SCC writer in straight-line code is an exec-mask operation (s_or_b64 exec / s_andn2_b64 exec / *_saveexec): selecting on
"exec is not empty after restoring it" is never what a program means.

    hipcc --offload-arch=gfx950 ... -S --cuda-device-only -o x.s x.hip;  python profiles/tools/scc_lint.py x.s ..."""
import re, sys
SCC_W = re.compile(r'^\s*(s_cmp_|s_cmpk_|s_bitcmp|s_and_b|s_or_b|s_xor_b|s_andn2_b|s_orn2_b|s_nand_b|s_nor_b|s_xnor_b|s_add_[iu]32|s_sub_[iu]32|'
                   r's_addc_u32|s_subb_u32|s_addk_i32|s_lshl_b|s_lshr_b|s_ashr_i|s_bfe_|s_min_|s_max_|s_not_b|s_abs_i32|s_absdiff|s_bcnt|'
                   r's_wqm|s_quadmask|s_and_saveexec|s_or_saveexec|s_xor_saveexec|s_andn2_saveexec|s_orn2_saveexec|s_andn1_saveexec|'
                   r's_lshl[1-4]_add|s_mul_i32x)')
SCC_R = re.compile(r'^\s*(s_cselect_b|s_cbranch_scc|s_cmov_b)')
EXEC_OP = re.compile(r'^\s*(s_(or|and|andn2|xor|orn2)_b64\s+exec\s*,|s_\w+_saveexec_b64)')
bad = 0
for fn in sys.argv[1:]:
    func, last = "?", None
    for ln, l in enumerate(open(fn), 1):
        if re.match(r'^[_A-Za-z]\S*:', l):
            func, last = l.split(':')[0], None
            continue
        if re.match(r'^\.LBB', l):
            last = None   # a join: unknown
            continue
        if SCC_R.match(l) and last is not None and EXEC_OP.match(last[1]):
            bad += 1
            print(f"{fn}:{ln}: {l.strip()}   <- SCC from {last[1].strip()} (line {last[0]}) in {func[:60]}")
        if SCC_W.match(l):
            last = (ln, l)
print(f"{bad} stale-SCC suspects in {len(sys.argv) - 1} file(s)")
sys.exit(1 if bad else 0)
