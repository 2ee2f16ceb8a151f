"""Static check of the hand-counted LDS waits in a hipcc -S listing.

The fused kernels read LDS through inline asm (ds_read_b128 / ds_read_b64_tr_b16) and wait with
asm `s_waitcnt lgkmcnt(N)`; the compiler does not know those registers are in flight.  This walks every
kernel linearly and reports any instruction that touches the destination of an asm LDS read before a
wait has covered it (LDS operations retire in order: a read is complete once a wait with
N <= number of LDS reads issued after it has executed).  Branch targets are ignored (straight-line model): the
kernels keep no LDS read in flight across a branch.  The asm global loads of dgrad's ReLU flags are checked the same
way against `s_waitcnt vmcnt(N)` while the code between load and use is straight-line; a use with a branch in
between is counted as not checkable (its wait relies on the loads a loop issues), not as a pass.

Second check, for the MFMAs issued from inline asm (the layer-pair weight-gradient kernel): the compiler's hazard
recogniser does not look into asm, so a VALU instruction that writes one of an asm MFMA's source registers must be at
least two wait states in front of it (an `s_nop N` counts N + 1; every other instruction counts one).

  python tools/check_lds_asm.py file.s   -> exit status 1 on a violation
"""
import re, sys

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")

def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out

unverified = [0]  # uses of an asm global load with a branch between load and use: outside the straight-line model
n_kernels = [0]   # kernels seen by check() so far: a listing in which none is recognised is an error, not a pass


def check(path):
    bad = 0
    kernel, pending, issued, in_asm = None, [], 0, False
    vpending = []   # asm global loads: [destination registers, line, vector-memory LOADS issued behind it]
    recent = []     # the last instructions of the kernel, youngest last: (line, code)
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"(_Z\w+):", s)          # (the label line carries a trailing `; @name` comment)
        if m:
            kernel, pending, issued, vpending, recent = m.group(1), [], 0, [], []
            n_kernels[0] += 1
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if s.startswith(";;#ASMEND"):
            in_asm = False; continue
        if not s or s.startswith((";", ".", "//")) or kernel is None:
            continue
        code = s.split(";")[0].strip()
        if in_asm and code.startswith("v_mfma"):
            srcs = set()
            for o in code.split(None, 1)[1].split(",")[1:]:
                srcs |= regs(o)
            ws = 0
            for l0, pc in reversed(recent):
                if ws >= 2:
                    break
                if pc.startswith("s_nop"):
                    ws += int(pc.split()[1]) + 1
                    continue
                if pc.startswith("v_") and not pc.startswith("v_mfma") and regs(pc.split(None, 1)[1].split(",")[0]) & srcs:
                    print(f"{path}:{ln}: {kernel[:60]}: asm `{code[:50]}` reads a register `{pc[:50]}` (line {l0}) wrote {ws} wait states earlier")
                    bad += 1
                ws += 1
        recent = (recent + [(ln, code)])[-4:]
        if in_asm and code.startswith("ds_read"):
            dst = regs(code.split(",")[0])
            issued += 1
            pending.append((issued, dst, ln))
            continue
        # asm global loads (dgrad's ReLU flags): loads retire in order, so one is complete once a
        # `s_waitcnt vmcnt(N)` executes with at least N loads issued behind it (stores sharing the counter
        # only make the wait stricter)
        if code.startswith(("global_load", "buffer_load", "scratch_load")):
            for v in vpending:
                v[2] += 1
            if in_asm and not code.startswith("global_load_lds"):
                vpending.append([regs(code.split(",")[0]), ln, 0, False])
                continue
        m = re.match(r"s_waitcnt\s+(.*)", code)
        if m:
            lg = re.search(r"lgkmcnt\((\d+)\)", m.group(1))
            if lg:
                n = int(lg.group(1))
                pending = [p for p in pending if issued - p[0] < n]   # younger than the n most recent stay
            vm = re.search(r"vmcnt\((\d+)\)", m.group(1))
            if vm:
                n = int(vm.group(1))
                vpending = [v for v in vpending if v[2] < n]
            continue
        if code.startswith("s_endpgm"):
            pending, vpending = [], []
            continue
        if code.startswith(("s_cbranch", "s_branch")):
            for v in vpending:
                v[3] = True
        touched = regs(code)
        for dst, l0, _, crossed in vpending:
            if touched & dst:
                if crossed:
                    # between load and use the listing is not straight-line code (a loop issues an unknown number of loads
                    # behind it, or layout order is not execution order): the count cannot be checked here
                    unverified[0] += 1
                else:
                    print(f"{path}:{ln}: {kernel[:60]}: `{code.strip()}` touches the destination of the global load at line {l0} before a wait covers it")
                    bad += 1
                break
        for seq, dst, l0 in pending:
            if touched & dst:
                print(f"{path}:{ln}: {kernel[:60]}: `{code.strip()}` touches the destination of the LDS read at line {l0} before its wait")
                bad += 1
                break
    return bad

if __name__ == "__main__":
    n = sum(check(p) for p in sys.argv[1:])
    print("violations:", n, "kernels checked:", n_kernels[0], "global-load uses behind control flow (not checkable):", unverified[0])
    sys.exit(1 if n or not n_kernels[0] else 0)
