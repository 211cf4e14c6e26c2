#!/usr/bin/env python3
"""Build-time check of the generated ISA of k_schur_slots and of the point-to-point back-substitution (run by the Makefile on mvba.s: the build FAILS when it fails).

The slot kernel keeps two gathers in flight with COUNTED `s_waitcnt vmcnt(N)`: every vector-memory
operation of its loops is inline assembly the compiler knows nothing about, and N is their number per iteration.
Correctness therefore rests on properties of the generated code that no C++ rule guarantees:
  1. the blocks of a loop (its own, wherever the compiler placed them) hold exactly N vector-memory operations -- the
     LDS-DMA gathers and the index row -- and no scratch / buffer access (a spill inside the loop would be an uncounted
     operation: the 64-bit-offset build of round 3 did exactly that and its results were wrong);
  2. no vector-memory operation in the loops returns data to a REGISTER (round 3's index loads did, into registers that
     had to be pinned: a value an asm load "returns" is not in its register yet, and a copy made before its counted
     wait once sent a gather to a stale address; since round 4 the indices travel through LDS like the records);
  3. M0 (the LDS-DMA destination base) is written by the gathers' own `s_mov_b32 m0, ...` only;
  4. the kernel fits three waves per SIMD (<= 168 VGPRs); registers it spills are touched outside the loops only
     (that is property 1).
Usage: check_isa.py mvba.s   (exit status 0 = all properties hold)."""
import re
import sys

# kernel -> (counted wait, LDS-DMA operations per iteration) of its diagonal / off-diagonal loop: the gathers + the indices
# (a step's indices are ONE 256-byte row; a kernel that adopts the loop -- the unit form tried it in round 4 -- adds a line)
KERNELS = {"k_schur_slots": (("vmcnt(7)", 7), ("vmcnt(8)", 8))}
VM_LOOP_OPS = ("global_load_lds_dwordx4", "global_load_lds_dword ")


def kernel_lines(text, name="k_schur_slots"):
    m = re.search(r"^_ZN\d+_GLOBAL__N_1" + str(len(name)) + name + r"E\w*:[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    if not m:
        return None
    return [ln.strip() for ln in m.group(1).splitlines()]


def loop_blocks(lines, wait_index):
    """Instructions of the Depth-1 blocks of the loop whose header holds the counted wait at `wait_index`, and those of its
    inner loops (the pacing block's poll loops).  The compiler annotates every block label -- `.LBBx_y: ; =>This [Inner] Loop
    Header: Depth=1` (followed by `;   Child Loop BBx_z Depth 2` lines), `;   in Loop: Header=BBx_y Depth=1` -- wherever it
    places the block (rotated loops keep their latch ABOVE the header, cold blocks go out of line)."""
    hdr = r"\.(LBB\w+):.*This (?:Inner )?Loop Header: Depth=1"
    head = max(i for i in range(wait_index + 1) if re.match(hdr, lines[i]))
    name = re.match(hdr, lines[head]).group(1)[1:]  # "BB5_35"
    children = set()
    for ln in lines[head + 1:]:
        m = re.match(r";\s+Child Loop (BB\w+) Depth", ln)
        if not m:
            break
        children.add(m.group(1))
    own, inner = [], []
    cur = None
    for i, ln in enumerate(lines):
        m = re.match(r"\.(LBB\w+):(.*)", ln)
        if m:
            label, note = m.group(1)[1:], m.group(2)
            if i == head or f"Header={name} Depth=1" in note:
                cur = own
            elif label in children or any(f"Header={c} " in note for c in children):
                cur = inner
            else:
                cur = None
            continue
        if cur is not None:
            cur.append(ln)
    return own, inner


def check_kernel(text, name, counts):
    errs = []
    lines = kernel_lines(text, name)
    if lines is None:
        return [f"{name} not found in the ISA"]
    for count, n_ops in counts:
        idx = [i for i, ln in enumerate(lines) if ln.startswith("s_waitcnt " + count)]
        if len(idx) != 1:
            errs.append(f"{name}: expected one `s_waitcnt {count}` (the loop's counted wait), found {len(idx)}")
            continue
        body, inner = loop_blocks(lines, idx[0])
        bad = [ln for ln in body + inner if ln.startswith(("scratch_", "buffer_"))]
        if bad:
            errs.append(f"{name} {count} loop: scratch / buffer access inside the loop: {bad[:3]}")
        # one iteration = the gathers + the indices, nothing else on the loop's own blocks (the slot form's pacing block --
        # poll and arrival -- are the inner loops: sc1 loads and atomics behind their own vmcnt(0))
        straight = [ln for ln in body if ln.startswith(VM_LOOP_OPS)]
        if len(straight) != n_ops:
            errs.append(f"{name} {count} loop: {len(straight)} LDS-DMA operations per iteration, the wait counts {n_ops}")
        other = [ln for ln in body if ln.startswith(("global_load", "global_store", "global_atomic", "flat_")) and not ln.startswith(VM_LOOP_OPS)]
        if other:
            errs.append(f"{name} {count} loop: vector-memory operations the wait does not count (or that return data to a register): {other[:3]}")
        stray = [ln for ln in inner if ln.startswith(VM_LOOP_OPS)]
        if stray:
            errs.append(f"{name} {count} loop: LDS-DMA inside the pacing block: {stray[:3]}")
    for ln in lines:
        if re.match(r"\w+\s+m0\b", ln) and not ln.startswith("s_mov_b32 m0,"):
            errs.append(f"{name}: M0 written outside the gathers' asm statements: {ln}")
    vg = re.search(r"^\s*\.set _ZN\d+_GLOBAL__N_1" + str(len(name)) + name + r"E\w*\.num_vgpr, (\d+)", text, re.M)
    if vg and int(vg.group(1)) > 168:
        errs.append(f"{name} needs {vg.group(1)} VGPRs: more than the 168 of three waves per SIMD")
    return errs


def check_flow_posts(text):
    """k_chol_backsolve_all<true> publishes y with sc1 stores and then a progress word (a 4-byte sc1 store by thread 0 behind
    a workgroup barrier).  Nothing in the language orders the two -- a workgroup-scope release fence is lowered to nothing in
    this execution mode -- so flow_post() carries an explicit `s_waitcnt vmcnt(0)`: every word store must find it between the
    last vector-memory operation before its barrier and that barrier.  (And the kernel must stay free of the device-wide
    fences it exists to avoid: buffer_wbl2 / buffer_inv.)"""
    m = re.search(r"^_ZN\d+_GLOBAL__N_1\d+k_chol_backsolve_allILb1EE\w*:[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    if not m:
        return ["k_chol_backsolve_all<true> not found in the ISA"]
    lines = [ln.strip() for ln in m.group(1).splitlines()]
    lines = [ln for ln in lines if ln and not ln.startswith(";")]
    errs = []
    words = [i for i, ln in enumerate(lines) if re.match(r"global_store_dword\s.*\bsc1\b", ln)]
    if not words:
        errs.append("k_chol_backsolve_all<true>: no progress-word store (global_store_dword ... sc1) found")
    for i in words:
        bars = [k for k in range(i) if lines[k].startswith("s_barrier")]
        if not bars:
            errs.append(f"k_chol_backsolve_all<true>: progress-word store without a barrier before it: {lines[i]}")
            continue
        ok = False
        for k in range(bars[-1] - 1, -1, -1):
            if re.match(r"s_waitcnt\b.*vmcnt\(0\)", lines[k]):
                ok = True
                break
            if lines[k].startswith(("global_", "buffer_", "flat_", "s_barrier")):
                break
        if not ok:
            errs.append(f"k_chol_backsolve_all<true>: a progress word may overtake the data it announces (no s_waitcnt vmcnt(0) "
                        f"between the last vector-memory operation and the barrier before `{lines[i]}`)")
    # the shared vector travels between workgroups through memory with sc1 loads and stores (a plain load could hit a stale line, a
    # plain store would sit in the L2): the annotations must still be there (which accesses are the vector's cannot be told from
    # the ISA -- the tiles of L are 8-byte loads too --, so this guards against their silent disappearance, not against a missing one)
    if not any(re.match(r"global_load_dwordx2\b.*\bsc1\b", ln) for ln in lines) or not any(re.match(r"global_store_dwordx2\b.*\bsc1\b", ln) for ln in lines):
        errs.append("k_chol_backsolve_all<true>: no sc1 load / store of the shared vector found")
    if any(ln.startswith(("buffer_wbl2", "buffer_inv")) for ln in lines):
        errs.append("k_chol_backsolve_all<true>: a device-wide fence (buffer_wbl2 / buffer_inv) crept back in")
    return errs


def check(text):
    errs = []
    for name, counts in KERNELS.items():
        errs += check_kernel(text, name, counts)
    errs += check_flow_posts(text)
    return errs


if __name__ == "__main__":
    errs = check(open(sys.argv[1] if len(sys.argv) > 1 else "mvba.s").read())
    for e in errs:
        print("check_isa: " + e, file=sys.stderr)
    if errs:
        sys.exit(1)
    print("check_isa: " + ", ".join(KERNELS) + " ok (counted waits, LDS-DMA only in the loops, M0, no scratch access in the loops); k_chol_backsolve_all<true> ok (stores complete before their progress word, no device-wide fence)")
