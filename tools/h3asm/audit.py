#!/usr/bin/env python
"""Audit of the compiled nsff_field_kernel_h3a (run by `make -C nsff_pl_amd/csrc audit`): the body between #ASMSTART / #ASMEND owns
v24..v255, a0..a255 and s40..s99 only WHILE IT RUNS -- hipcc may use them before and after (the clobber list tells it that nothing
of its own survives the statement).  What must hold: no spilled VGPRs, no scratch (scalars parked in lanes of a VGPR the
body leaves alone -- the persistent tile loop's invariants -- are reported, not refused), 512 registers allocated, one asm statement,
and between the kernel's entry and the asm no compiler code may leave a value in the asm-owned range that it reads back after
the asm (the clobber list guarantees it; the audit reports the count of compiler instructions and the descriptor fields).
No compiler instruction of the kernel may touch an accumulation register: the first eight weight slots are in flight or resident
in front of the body and (a persistent workgroup's tile loop: requested by the previous tile's B16LP phase) behind it."""
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "nsff_pl_amd/csrc/field_h3.hip"
out = "/tmp/field_h3_audit.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                      stderr=subprocess.DEVNULL)
txt = open(out).read()
failed = []
for kernel in ("nsff_field_kernel_h3a", "nsff_field_kernel_h3a_save"):          # inference body, training-forward (SAVE) body
  m = re.search(r'^(_ZN\S*' + kernel + r'E\S*):', txt, re.M)
  name = m.group(1)
  body = txt[m.end():]
  body = body[:body.index(".Lfunc_end")]
  lines = body.split("\n")
  # the body is the one asm statement that holds MFMAs (the encoder / heads have one-instruction statements of their own); the
  # pre-issue statements are the ones with global loads into accumulation registers (one weight slot each, spread over the
  # encoder) -- from the first of them to the body the compiler's code (the input encoder) must not touch an accumulation
  # register: the loads are in flight
  n_asm, n_pre, n_pre_loads, inasm, before, after, seen, cur, curl = 0, 0, 0, False, 0, 0, False, 0, 0
  between, agpr_between, slots_seen = True, [], set()      # (the whole kernel: a persistent workgroup's loop top runs with slots 0..7 resident too)
  for ln in lines:
      if "#ASMSTART" in ln:
          inasm, cur, curl, cur_slots = True, 0, 0, set()
          continue
      if "#ASMEND" in ln:
          inasm = False
          if cur > 100:
              # (the tile loop of a persistent workgroup: the body's B16LP phase leaves the NEXT tile's weight slots 0..7 resident --
              #  the compiler's code behind the body and up to the loop's back edge is held to the same rule)
              n_asm, seen = n_asm + 1, True
          elif curl >= 4:
              n_pre, n_pre_loads, between = n_pre + 1, n_pre_loads + curl, True
              slots_seen |= cur_slots
          continue
      t = ln.strip()
      if inasm:
          cur += "v_mfma" in t
          if t.startswith("global_load_dwordx4 a["):
              curl += 1
              cur_slots.add(int(t.split("a[")[1].split(":")[0]) // 16)
          continue
      if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
          continue
      if between and (re.search(r"\ba\[?\d", t.split(";")[0]) or "accvgpr" in t):
          agpr_between.append(t)
      if seen:
          after += 1
      else:
          before += 1
  k = txt.index(".amdhsa_kernel " + name)
  meta = txt[k:k + 4000]
  get = lambda key: re.search(r"\.amdhsa_" + key + r"\s+(\S+)", meta).group(1)
  md = txt[txt.index("amdhsa.kernels"):]
  md = md[md.index(name):]
  spill_v = int(re.search(r"\.vgpr_spill_count:\s*(\d+)", md).group(1))
  spill_s = int(re.search(r"\.sgpr_spill_count:\s*(\d+)", md).group(1))
  print(f"{name}: asm statements {n_asm}, compiler instructions before / after the body {before} / {after}")
  print(f"  next_free_vgpr {get('next_free_vgpr')}  accum_offset {get('accum_offset')}  next_free_sgpr {get('next_free_sgpr')}  "
        f"scratch {get('private_segment_fixed_size')} B  LDS {get('group_segment_fixed_size')} B  spills: {spill_v} VGPR, {spill_s} SGPR")
  print(f"  pre-issue statements {n_pre} ({n_pre_loads} loads, one weight slot each, slots {sorted(slots_seen)}: three behind the workgroup's own "
        f"loads, five inside either encoder); compiler instructions touching accumulation registers anywhere in the kernel: {len(agpr_between)}")
  for t in agpr_between[:8]:
      print("     ", t)
  ok = slots_seen == set(range(8)) and n_pre_loads == 4 * n_pre and not agpr_between and n_asm == 1 and spill_v == 0 and int(get("private_segment_fixed_size")) == 0 and int(get("next_free_vgpr")) == 512
  if not ok:
      failed.append(kernel)
sys.exit(0 if not failed else f"AUDIT FAILED: {failed}")
