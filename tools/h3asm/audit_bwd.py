#!/usr/bin/env python
"""Audit of the compiled nsff_field_bwd_kernel_h3b (run by `make -C nsff_pl_amd/csrc audit`): the body between #ASMSTART / #ASMEND
owns v24..v255, a0..a127 and s40..s99 while it runs.  What must hold: one asm statement with the body's MFMAs, no spilled VGPRs,
no scratch, 512 registers per lane available (one wave per SIMD: launch_bounds(256, 1)), the LDS image the body addresses
(gradient tile + stash tile + scales + the next item's records) within 160 KB.  The workgroup is PERSISTENT: the statement sits in
a loop over the workgroup's items, and what the compiler keeps across it (the item bookkeeping, the thread index, kernel
arguments) can only live in v0..v23, s0..s39 -- the statement's clobber list says so -- or in lanes of such a VGPR (SGPR spills:
allowed; VGPR spills and scratch: not)."""
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "nsff_pl_amd/csrc/field_bwd.hip"
out = "/tmp/field_bwd_audit.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                      stderr=subprocess.DEVNULL)
txt = open(out).read()
kernel = "nsff_field_bwd_kernel_h3b"
m = re.search(r'^(_ZN\S*' + kernel + r'E\S*):', txt, re.M)
name = m.group(1)
body = txt[m.end():]
body = body[:body.index(".Lfunc_end")]
n_asm, inasm, cur, before, after, seen, n_mfma, first_after = 0, False, 0, 0, 0, False, 0, ""
n_pre, pre_loads, between, agpr_between, curl = 0, 0, False, [], 0
for ln in body.split("\n"):
    if "#ASMSTART" in ln:
        inasm, cur, curl = True, 0, 0
        continue
    if "#ASMEND" in ln:
        inasm = False
        if cur > 100:
            n_asm, seen, n_mfma = n_asm + 1, True, cur
        elif curl >= 4:
            n_pre, pre_loads, between = n_pre + 1, pre_loads + curl, True
        continue
    t = ln.strip()
    if inasm:
        cur += "v_mfma" in t
        curl += t.startswith("global_load_dwordx4 a[")
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    if between and not seen and (re.search(r"\ba\[?\d", t.split(";")[0]) or "accvgpr" in t):
        agpr_between.append(t)
    if seen:
        after += 1
        if after == 1:
            first_after = t
    else:
        before += 1
k = txt.index(".amdhsa_kernel " + name)
meta = txt[k:k + 4000]
get = lambda key: re.search(r"\.amdhsa_" + key + r"\s+(\S+)", meta).group(1)
md = txt[txt.index("amdhsa.kernels"):]
md = md[md.index(name):]
spill_v = int(re.search(r"\.vgpr_spill_count:\s*(\d+)", md).group(1))
spill_s = int(re.search(r"\.sgpr_spill_count:\s*(\d+)", md).group(1))
print(f"{name}: asm statements {n_asm} ({n_mfma} MFMAs in the body), compiler instructions in front of the body {before}, first instruction behind it: {first_after}")
print(f"  next_free_vgpr {get('next_free_vgpr')}  accum_offset {get('accum_offset')}  next_free_sgpr {get('next_free_sgpr')}  "
      f"scratch {get('private_segment_fixed_size')} B  LDS {get('group_segment_fixed_size')} B  spills: {spill_v} VGPR, {spill_s} SGPR")
print(f"  pre-issue statements {n_pre} ({pre_loads} weight-slot loads in flight across the head stage); compiler instructions touching "
      f"accumulation registers between it and the body: {len(agpr_between)}")
ok = (n_asm == 1 and n_pre == 1 and pre_loads == 8 and not agpr_between and spill_v == 0 and int(get("private_segment_fixed_size")) == 0 and int(get("next_free_vgpr")) >= 384
      and int(get("group_segment_fixed_size")) <= 160 * 1024)
sys.exit(0 if ok else "AUDIT FAILED: " + kernel)
