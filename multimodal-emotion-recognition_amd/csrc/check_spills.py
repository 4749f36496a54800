#!/usr/bin/env python3
"""Build-time guard for gemm.hip (run by the Makefile on hipcc's -Rpass-analysis=kernel-resource-usage remarks).

The k-contiguous staging ring issues its global loads from inline asm and waits with hand-counted vmcnt
(Stage16KC::wait_loaded): the compiler does not know those registers are in flight, so it must never SPILL or move them
between issue and wait - a spilled register is reused for addresses and then overwritten by the landing load (a
512-thread, 128-VGPR build with three register sets did exactly that and faulted).  Every kernel of that form
(m2f_gemm16*<A_RC=false, B_RC=false, ...>, mangled `ILb0ELb0E`) must therefore have zero VGPR spills."""
import re
import sys

# --ring <remarks>: the ring-form kernels (gemm_ring.h) wait for their LDS-DMA loads with hand-counted `s_waitcnt vmcnt(N)`
# (ring_wait_vm): scratch loads / stores are vector-memory operations on the same counter, so ANY scratch use (spills) in
# such a kernel makes the counts wrong - the build must not produce one.
ring = len(sys.argv) > 2 and sys.argv[1] == "--ring"
path = sys.argv[2] if ring else sys.argv[1]
rows, cur = [], None
for line in open(path, errors="replace"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
    elif cur is not None and "remark" in line:
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m:
            cur["vgpr_spill"] = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m:
            cur["scratch"] = int(m.group(1))
if ring:
    kernels = [r for r in rows if "m2f_gemm16_ring_kernel" in r["name"] or "m2f_gemm_p8_kernel" in r["name"]]
    if not kernels:
        sys.exit(f"check_spills: no ring kernel found in {path} - did the remark format change?")
    bad = [r for r in kernels if r.get("scratch", 0) > 0 or r.get("vgpr_spill", 0) > 0]
    for r in bad:
        print(f"check_spills: {r['name']} uses {r.get('scratch', 0)} bytes of scratch per lane ({r.get('vgpr_spill', 0)} VGPRs spilled) "
              "but counts its LDS-DMA loads by hand", file=sys.stderr)
    sys.exit(1 if bad else 0)
bad = [r for r in rows if "m2f_gemm16" in r["name"] and "ILb0ELb0E" in r["name"] and r.get("vgpr_spill", 0) > 0]
checked = sum(1 for r in rows if "m2f_gemm16" in r["name"] and "ILb0ELb0E" in r["name"])
if checked == 0:
    sys.exit("check_spills: no m2f_gemm16 NT kernel found in the remarks - did the remark format change?")
for r in bad:
    print(f"check_spills: {r['name']} spills {r['vgpr_spill']} VGPRs but stages with inline-asm loads", file=sys.stderr)
sys.exit(1 if bad else 0)
