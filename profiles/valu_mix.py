#!/usr/bin/env python3
"""
Static instruction mix of the sweep kernels, from their ISA (make -C fcdiff_amd/csrc fcd_gibbs.s fcd_gibbs_r.s):

    python profiles/valu_mix.py > profiles/r04_valu_mix.json

For each kernel: the number of vector-ALU instructions in its text and the share of the "4-cycle class" -- what
profiles/r03_ubench_valu_rate.txt measured at 2.44 issue units against 1.42 for plain two-source 32-bit work: every fp64
op, packed fp32, three-source integer ops (v_and_or, v_lshl_add, v_bfe, v_perm, v_mad_*), v_mul_lo/hi, compares,
v_cndmask, conversions, transcendentals.  A STATIC share (the text, not the executed stream: the unrolled hot loops
dominate both); bench.py uses it to price SQ_INSTS_VALU in `roofline_valu`.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"gibbs_f_pair_kernel": ("fcd_gibbs.s", "_ZN12_GLOBAL__N_119gibbs_f_pair_kernelILi4E"),
           "gibbs_f_pairx_kernel": ("fcd_gibbs.s", "_ZN12_GLOBAL__N_120gibbs_f_pairx_kernelILi4E"),
           "gibbs_r_pipe_kernel": ("fcd_gibbs_r.s", "_ZN12_GLOBAL__N_119gibbs_r_pipe_kernelILi2ELi8E"),
           "gibbs_r_step_kernel": ("fcd_gibbs_r.s", "_ZN12_GLOBAL__N_119gibbs_r_step_kernelILi1ELi8E")}
WIDE = re.compile(r"^v_(\w*_f64|pk_\w+|and_or_b32|or3_b32|lshl_add_u32|lshl_or_b32|add3_u32|bfe_[ui]32|perm_b32|mad_\w+|mul_lo_u32|mul_hi_u32|"
                  r"cmp\w*|cndmask_b32|cvt_\w+|exp_f32|log_f32|rcp_\w+|sqrt_\w+|readlane_b32|writelane_b32|lshl_add_u64|max3_\w+|min3_\w+|fma_\w+)")


def main():
    out = {}
    for (k, (fname, sym)) in KERNELS.items():
        path = os.path.join(ROOT, "fcdiff_amd", "csrc", fname)
        if not os.path.exists(path):
            continue
        text = subprocess.run([os.path.join(ROOT, "profiles", "kernel_asm.sh"), path, sym], capture_output=True, text=True).stdout
        n = wide = 0
        for line in text.splitlines():
            t = line.strip()
            if not t.startswith("v_") or t.startswith("v_nop"):
                continue
            n += 1
            if WIDE.match(t):
                wide += 1
        if n:
            out[k] = {"valu_instructions_in_text": n, "frac_4_cycle": wide / n, "symbol": sym}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
