"""Build-time audit of the eight-phase and ring GEMM kernels in mraudio_amd/csrc/gemm.hip (run by the Makefile before gemm.o is built).

``gemm_p8_tile`` counts its own vector-memory queue: per pair of K tiles eight LDS-DMA half-tiles are issued and two counted
``s_waitcnt vmcnt(6)`` (``vmcnt(9)`` on the 128 x 512 tail tile) leave exactly the three youngest in flight.  That arithmetic only
holds if the compiler adds nothing to the queue and waits on nothing by itself inside the loop: a spill (scratch store / load), an
ordinary global load it sank into the loop, or a conservative ``s_waitcnt vmcnt(0)`` would either break the count or drain the
prefetch and silently cost the overlap (cdna_hip_programming.md section 5.7).  A different compile can change any of that, so the
check runs on every build of the shipped object.  For every ``gemm_p8_kernel`` / ``gemm_p8_mixed_kernel`` instantiation, inside
each K loop (an innermost loop that holds MFMAs and barriers):

  * no scratch access;
  * no vector-memory instruction other than the LDS-DMA ``global_load_lds_dwordx4``;
  * every ``s_waitcnt`` with a ``vmcnt`` field comes from the source's own ``asm volatile`` (between ASMSTART / ASMEND markers) and
    is one of vmcnt(6) / vmcnt(9) (steady state) or vmcnt(0) (the last pair of K tiles);
  * the loop body is the fully unrolled pair of K tiles: 16 barriers, 128 MFMAs, 16 LDS-DMA instructions (20 on the tail tile: 2 x (1 + 1 + 4 + 4)).

    python tools/audit_gemm_p8.py build/obj/gemm.gfx950.s
"""
import re
import sys


def split_blocks(lines):
    """[(label, comment, first, last)] of the basic blocks of one kernel."""
    blocks, cur = [], ("entry", "", 0)
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l.strip())
        if m:
            blocks.append((cur[0], cur[1], cur[2], i - 1))
            cur = (m.group(1), m.group(2) or "", i)
    blocks.append((cur[0], cur[1], cur[2], len(lines) - 1))
    return blocks


def k_loops(lines):
    """Line-index sets of the innermost loops that contain MFMAs and barriers."""
    loops = {}
    for label, comment, a, b in split_blocks(lines):
        hdr = None
        if "Inner Loop Header" in comment:
            hdr = label[1:]                       # ".LBB18_20" -> "LBB18_20"
        else:
            m = re.search(r"in Loop: Header=(BB\d+_\d+)", comment)
            if m:
                hdr = "L" + m.group(1)
        if hdr:
            loops.setdefault(hdr, []).extend(range(a, b + 1))
    out = []
    for hdr, idx in loops.items():
        text = [lines[i] for i in idx]
        if any("v_mfma" in t for t in text) and any(t.strip().startswith("s_barrier") for t in text):
            out.append((hdr, sorted(idx)))
    return out


def audit(name, lines):
    errs = []
    tail = "Lb1EEE" in name            # gemm_p8_kernel<T, EPI, TAIL = true>
    mixed = "gemm_p8_mixed_kernel" in name
    loops = k_loops(lines)
    want = 2 if mixed else 1           # the mixed kernel holds both tile forms
    if len(loops) != want:
        errs.append(f"expected {want} K loop(s), found {len(loops)}")
    for hdr, idx in loops:
        in_asm = False
        nbar = nmfma = ndma = 0
        seen_wait = []
        first = idx[0]
        # ASMSTART state must be tracked over the whole kernel text up to each line
        state = {}
        cur = False
        for i, l in enumerate(lines):
            s = l.strip()
            if s.startswith(";;#ASMSTART"):
                cur = True
            elif s.startswith(";;#ASMEND"):
                cur = False
            state[i] = cur
        for i in idx:
            s = lines[i].strip()
            if not s or s.startswith((";", ".")) or s.endswith(":"):
                continue
            s = s.split(";")[0].strip()
            if not s:
                continue
            op = s.split()[0]
            if op.startswith("scratch_"):
                errs.append(f"{hdr} line {i}: scratch access inside the K loop: '{s}'")
            elif op == "global_load_lds_dwordx4":
                ndma += 1
            elif op.startswith(("global_", "buffer_", "flat_")):
                errs.append(f"{hdr} line {i}: vector-memory instruction other than the LDS-DMA inside the K loop: '{s}'")
            elif op == "s_barrier":
                nbar += 1
            elif op.startswith("v_mfma"):
                nmfma += 1
            elif op == "s_waitcnt" and "vmcnt" in s:
                n = int(re.search(r"vmcnt\((\d+)\)", s).group(1))
                seen_wait.append(n)
                if not state[i]:
                    errs.append(f"{hdr} line {i}: compiler-inserted '{s}' inside the K loop")
        big = ndma > 16
        steady = 9 if big else 6
        if sorted(set(seen_wait)) not in ([0, steady], [steady]):
            errs.append(f"{hdr}: vmcnt waits in the loop are {sorted(seen_wait)}, expected two vmcnt({steady}) and one vmcnt(0)")
        if seen_wait.count(steady) != 2 or seen_wait.count(0) > 1:
            errs.append(f"{hdr}: {seen_wait.count(steady)} x vmcnt({steady}), {seen_wait.count(0)} x vmcnt(0); expected 2 and <= 1")
        if nbar != 16 or nmfma != 128 or ndma not in (16, 20):
            errs.append(f"{hdr}: {nbar} barriers, {nmfma} MFMAs, {ndma} LDS-DMA instructions; expected 16 / 128 / 16 (20 on the tail tile)")
        if not mixed and (ndma == 20) != tail:
            errs.append(f"{hdr}: LDS-DMA count {ndma} does not match the tile form (tail = {tail})")
    return errs


def audit_ring(name, lines):
    """gemm_ring_kernel counts its LDS-DMA queue too ((STAGES - 3) x pieces per wave stay in flight): inside its K loop no scratch access, no
    vector-memory instruction other than the LDS-DMA, and no ``s_waitcnt vmcnt`` that is not the source's own."""
    errs = []
    loops = k_loops(lines)
    if not loops:
        errs.append("K loop not found")
    state, cur = {}, False
    for i, l in enumerate(lines):
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            cur = True
        elif s.startswith(";;#ASMEND"):
            cur = False
        state[i] = cur
    for hdr, idx in loops:
        ndma = 0
        for i in idx:
            s = lines[i].strip().split(";")[0].strip()
            if not s or s.startswith(".") or s.endswith(":"):
                continue
            op = s.split()[0]
            if op.startswith("scratch_"):
                errs.append(f"{hdr} line {i}: scratch access inside the K loop: '{s}'")
            elif op == "global_load_lds_dwordx4":
                ndma += 1
            elif op.startswith(("global_", "buffer_", "flat_")):
                errs.append(f"{hdr} line {i}: vector-memory instruction other than the LDS-DMA inside the K loop: '{s}'")
            elif op == "s_waitcnt" and "vmcnt" in s and not state[i]:
                errs.append(f"{hdr} line {i}: compiler-inserted '{s}' inside the K loop")
        if ndma == 0:
            errs.append(f"{hdr}: no LDS-DMA inside the K loop")
    return errs


def main(path):
    text = open(path).read().splitlines()
    kernels, cur = {}, None
    for l in text:
        m = re.match(r"^(_ZN3mra[^:]*gemm_(?:p8_(?:mixed_)?|ring_)kernel[^:]*):", l)
        if m:
            cur = []
            kernels[m.group(1)] = cur
        elif cur is not None:
            cur.append(l)
            if l.strip().startswith(".Lfunc_end"):     # not s_endpgm: the mixed kernel returns early for an odd number of row tiles
                cur = None
    if len(kernels) < 6:
        print(f"audit_gemm_p8: only {len(kernels)} eight-phase kernels found in", path)
        return 1
    bad = 0
    for name, lines in kernels.items():
        errs = audit_ring(name, lines) if "gemm_ring_kernel" in name else audit(name, lines)
        print(f"audit_gemm_p8: {name}: {'OK' if not errs else 'FAILED'}")
        for e in errs[:12]:
            print("   ", e)
        bad += bool(errs)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
