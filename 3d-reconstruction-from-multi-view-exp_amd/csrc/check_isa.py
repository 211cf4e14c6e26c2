#!/usr/bin/env python3
"""Build-time check of the generated ISA of k_schur_slots (run by the Makefile on mvba.s: the build FAILS when it fails).

The slot-resident Schur kernel keeps two gathers in flight with COUNTED `s_waitcnt vmcnt(N)`: every vector-memory
operation of its loops is inline assembly the compiler knows nothing about, and N is their number per iteration.
Correctness therefore rests on properties of the generated code that no C++ rule guarantees:
  1. between the two counted waits of a loop (unrolled by two) there are exactly N vector-memory operations and no
     scratch / buffer access (a spill inside the loop would be an uncounted operation: the 64-bit-offset build did
     exactly that and its results were wrong);
  2. the index registers an asm load fills are pinned (v152..v167): only those loads (and the zeros that initialise
     them) write one, and no move reads one (a copy made between a load and its counted wait reads the register before
     the data has landed: seen once, the gather went to a stale address);
  3. M0 (the LDS-DMA destination base) is written by the gathers' own `s_mov_b32 m0, ...` only;
  4. the kernel fits three waves per SIMD (<= 168 VGPRs); registers it spills are touched outside the loops only
     (that is property 1).
Usage: check_isa.py mvba.s   (exit status 0 = all properties hold)."""
import re
import sys

PINNED = r"v1(?:5[2-9]|6[0-7])\b"
COUNTS = (("vmcnt(12)", 12), ("vmcnt(14)", 14))  # diagonal / off-diagonal loop: operations per iteration
VM_LOOP_OPS = ("global_load_lds_dwordx4", "global_load_dword ")


def kernel_lines(text, name="k_schur_slots"):
    m = re.search(r"^_ZN\d+_GLOBAL__N_1\d+" + name + r"E\w*:[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    if not m:
        return None
    return [ln.strip() for ln in m.group(1).splitlines()]


def check(text):
    errs = []
    lines = kernel_lines(text)
    if lines is None:
        return ["k_schur_slots not found in the ISA"]
    for count, n_ops in COUNTS:
        idx = [i for i, ln in enumerate(lines) if ln.startswith("s_waitcnt " + count)]
        if len(idx) != 2:  # the loop is unrolled by two
            errs.append(f"expected two `s_waitcnt {count}` (loop unrolled by two), found {len(idx)}")
            continue
        body = lines[idx[0]:idx[1]]
        bad = [ln for ln in body if ln.startswith(("scratch_", "buffer_"))]
        if bad:
            errs.append(f"{count} loop: scratch / buffer access between the counted waits: {bad[:3]}")
        # one iteration = the gathers + the index loads, nothing else on the straight path (the pacing block's poll and
        # arrival sit behind a branch that is not taken between segment boundaries: sc1 loads / atomics)
        straight = [ln for ln in body if ln.startswith(VM_LOOP_OPS) and "sc1" not in ln]
        if len(straight) != n_ops:
            errs.append(f"{count} loop: {len(straight)} vector-memory operations per iteration, the wait counts {n_ops}")
        other = [ln for ln in body if ln.startswith(("global_load", "global_store", "flat_")) and not ln.startswith(VM_LOOP_OPS)
                 and "sc1" not in ln]
        if other:
            errs.append(f"{count} loop: uncounted vector-memory operations: {other[:3]}")
        start = max([i for i in range(idx[0]) if lines[i].startswith(("global_store", "global_atomic", "s_endpgm"))] or [0])
        region = lines[start:idx[1]]  # prologue + loop (from the end of whatever wrote results before)
        idx_loads = [ln for ln in region if ln.startswith("global_load_dword ") and "sc1" not in ln]
        stray = [ln for ln in idx_loads if not re.match(r"global_load_dword " + PINNED, ln)]
        if len(idx_loads) < 24 or stray:
            errs.append(f"{count} form: index loads outside the pinned registers v152..v167: {stray[:3]} ({len(idx_loads)} loads)")
    for ln in lines:
        w = re.match(r"(\w+)\s+(?:v\[)?" + PINNED, ln)
        if w and not ln.startswith("global_load_dword ") and not re.match(r"v_mov_b32_e32 " + PINNED + r", 0$", ln):
            errs.append(f"a pinned index register is written by something other than its load: {ln}")
        if ln.startswith("v_mov_b32") and re.search(r", " + PINNED, ln):
            errs.append(f"a move reads a pinned index register: {ln}")
        if re.match(r"\w+\s+m0\b", ln) and not ln.startswith("s_mov_b32 m0,"):
            errs.append(f"M0 written outside the gathers' asm statements: {ln}")
    vg = re.search(r"^\s*\.set _ZN\d+_GLOBAL__N_113k_schur_slotsE\w*\.num_vgpr, (\d+)", text, re.M)
    if vg and int(vg.group(1)) > 168:
        errs.append(f"k_schur_slots needs {vg.group(1)} VGPRs: more than the 168 of three waves per SIMD")
    return errs


if __name__ == "__main__":
    errs = check(open(sys.argv[1] if len(sys.argv) > 1 else "mvba.s").read())
    for e in errs:
        print("check_isa: " + e, file=sys.stderr)
    if errs:
        sys.exit(1)
    print("check_isa: k_schur_slots ok (counted waits, pinned index registers, M0, no scratch access in the loops)")
