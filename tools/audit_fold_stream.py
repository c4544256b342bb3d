"""Build-time audit of mraudio_amd/csrc/fold_stream.hip's generated assembly (run by the Makefile).

The streaming kernels keep their row operand in flight in ordinary VGPRs that inline-assembly loads write two K steps
before the counted ``s_waitcnt vmcnt`` that covers them.  hipcc does not know that: if it spilled, copied or reused one
of those registers between the load and the wait, the kernel would read garbage without any fault
(/opt/skills/guides/cdna_hip_programming.md section 5.7 item 1).  This script fails the build unless, for every
``fold_stream_kernel`` instantiation:

  * there is no scratch use at all (no spill can touch an in-flight register, and no compiler ``s_waitcnt vmcnt`` for a
    scratch access can drain the hand-counted queue);
  * no instruction reads or writes a register whose assembly ``global_load_dwordx4`` has not yet been retired by one
    of the assembly's counted ``s_waitcnt vmcnt(N)`` (the queue is replayed in issue order, the K loop twice);
  * the compiler emitted no ``s_waitcnt vmcnt`` of its own inside the K loop.

    python tools/audit_fold_stream.py build/obj/fold_stream-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def audit(name, lines):
    errs = []
    if any("scratch_" in l for l in lines):
        errs.append("scratch access (register spill) present")
    bars = [i for i, l in enumerate(lines) if l.strip().startswith("s_barrier")]
    if len(bars) < 2:
        return errs + ["K loop not found (fewer than two s_barrier)"]

    # The vector-memory queue in issue order: each assembly global_load_dwordx4 with the registers it will write, each
    # LDS-DMA with none.  An assembly s_waitcnt vmcnt(N) retires all but the N youngest.  Until its load has retired, a
    # register may be touched by NOTHING (the compiler believes it was written when the load was issued).
    def scan(idx_range, fifo):
        in_asm = False
        for i in idx_range:
            s = lines[i].strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith((";", ".")) or s.endswith(":"):
                continue
            s = s.split(";")[0].strip()
            op = s.split()[0]
            toks = [t.rstrip(",") for t in s.split()[1:]]
            if op == "s_waitcnt" and "vmcnt" in s:
                n = int(re.search(r"vmcnt\((\d+)\)", s).group(1))
                if not in_asm and bars[0] <= i <= bars[-1]:
                    errs.append(f"line {i}: compiler-inserted '{s}' inside the K loop")
                del fifo[: max(0, len(fifo) - n)]
                continue
            pending = set().union(*fifo) if fifo else set()
            touched = set()
            for t in toks:
                touched |= regs_of(t)
            if touched & pending:
                errs.append(f"line {i}: '{s}' touches registers {sorted(touched & pending)[:4]} whose load is still in flight")
            if in_asm and op == "global_load_dwordx4":
                fifo.append(regs_of(toks[0]))
            elif in_asm and op.startswith(("buffer_load", "global_load_lds")):
                fifo.append(set())
            elif not in_asm and op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                fifo.append(set())      # a compiler-issued vector-memory operation also sits in the queue
        return fifo

    scan(range(len(lines)), [])
    # steady state: the K loop (the backward branch around the first barrier) once more, entered with the queue it leaves behind
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l.strip())
        if m:
            labels[m.group(1)] = i
    loop = None
    for i, l in enumerate(lines):
        m = re.match(r"^\s*s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < bars[0] < i:
            loop = (labels[m.group(1)], i) if loop is None or i - labels[m.group(1)] < loop[1] - loop[0] else loop
    if loop is None:
        errs.append("K loop (backward branch around the first barrier) not found")
    else:
        scan(range(loop[0], loop[1] + 1), scan(range(0, loop[1] + 1), []))
    if not any(lines[i].strip().startswith("global_load_dwordx4") for i in range(len(lines))):
        errs.append("no assembly global_load_dwordx4 found")
    seen, out = set(), []
    for e in errs:
        if e not in seen:
            seen.add(e)
            out.append(e)
    return out


def main(path):
    text = open(path).read().splitlines()
    kernels, cur, name = {}, None, None
    for l in text:
        m = re.match(r"^(_ZN3mra[^:]*fold_stream_kernel[^:]*):", l)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
        elif cur is not None:
            cur.append(l)
            if l.strip().startswith("s_endpgm"):
                cur = None
    if not kernels:
        print("audit_fold_stream: no fold_stream_kernel found in", path)
        return 1
    bad = 0
    for name, lines in kernels.items():
        errs = audit(name, lines)
        print(f"audit_fold_stream: {name}: {'OK' if not errs else 'FAILED'}")
        for e in errs[:12]:
            print("   ", e)
        bad += bool(errs)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
