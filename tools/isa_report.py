#!/usr/bin/env python3
"""Register / scratch / memory-instruction report of the gfx950 ISA of selected kernels.

  tools/isa_report.py <file.s> <regex on the mangled name>

<file.s> from:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S x.hip -o x.s
(The kernel-trace column VGPR_Count of rocprofv3 is NOT the allocation; this is.)"""
import re
import sys


def main():
    src = open(sys.argv[1]).read()
    pat = re.compile(sys.argv[2])
    for m in re.finditer(r'^(_Z\w+):\s*;[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel', src, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not pat.search(name):
            continue
        def f(key):
            r = re.search(rf'\.amdhsa_{key} (\d+)', body)
            return int(r.group(1)) if r else None
        code = body.split(".amdhsa_kernel")[0]
        waits = re.findall(r's_waitcnt vmcnt\((\d+)\)', code)
        print(f"{name}\n   vgpr {f('next_free_vgpr')}  accum_offset {f('accum_offset')}  sgpr {f('next_free_sgpr')}  "
              f"scratch {f('private_segment_fixed_size')} B  loads x4 {len(re.findall(r'global_load_dwordx4', code))}  "
              f"stores x4 {len(re.findall(r'global_store_dwordx4', code))}  v_accvgpr {len(re.findall(r'v_accvgpr', code))}  "
              f"scratch_ops {len(re.findall(r'scratch_(load|store)', code))}")
        print(f"   s_waitcnt vmcnt(N) in program order: {' '.join(waits)}")


if __name__ == "__main__":
    main()
