#!/usr/bin/env python
"""Where do a kernel's register spills execute?
    tools/check_spills.py file.gfx950.s [more.gfx950.s ...] [kernel-name-substring ...] [--no-scratch name ...] [--max-inner name=N ...]

For every kernel of the listing that has spills (scratch_load / scratch_store = VGPR spills, v_writelane_b32 / v_readlane_b32 with
the compiler's "SGPR spill" comment = SGPR spills to VGPR lanes) this prints how many of those instructions sit inside a LOOP
THAT CONTAINS MFMAs (a hot loop: the range between a label and a later backward branch to it), and exits non-zero if any
does for the kernels named on the command line.  Spills outside every such loop run once per workgroup (prologue / epilogue) or
once per job, not per tile ("show they are off every path that runs per tile").

A name that matches NO kernel of the listings given is an error (the names are mangled substrings, and a renamed
template would otherwise leave the gate passing on nothing).  `--max-inner name=N`: the kernel may hold up to N spill operations
inside an innermost MFMA loop (the fp32 parity kernels, whose only MFMA loop is the pass loop: a stated budget, so that a
regression shows)."""
import re
import sys


def kernels(txt):
    for m in re.finditer(r"^(_Z\w+):.*?^\.Lfunc_end\d+:", txt, re.S | re.M):
        yield m.group(1), m.group(0)


def analyse(body):
    lines = [l.strip() for l in body.split("\n")]
    label_at = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in label_at and label_at[tgt] < i:
                loops.append((label_at[tgt], i))
    hot = [(a, b) for a, b in loops if any(x.startswith("v_mfma") for x in lines[a:b])]
    vg, sg, vg_hot, sg_hot = 0, 0, [], []
    for i, l in enumerate(lines):
        is_v = l.startswith("scratch_load") or l.startswith("scratch_store")
        # (hipcc writes an SGPR spill as v_writelane_b32 / v_readlane_b32 of a reserved VGPR, without a comment; these kernels use
        #  neither instruction for anything else)
        is_s = l.startswith("v_writelane_b32") or l.startswith("v_readlane_b32")
        if not (is_v or is_s):
            continue
        inside = [(a, b) for a, b in hot if a <= i <= b]
        if is_v:
            vg += 1
            if inside:
                vg_hot.append((i, l, min(b - a for a, b in inside)))
        else:
            sg += 1
            if inside:
                sg_hot.append((i, l, min(b - a for a, b in inside)))
    return len(hot), vg, sg, vg_hot, sg_hot, lines, hot


def main():
    args = sys.argv[1:]
    files = [a for a in args if a.endswith(".s")]
    args = [a for a in args if not a.endswith(".s")]
    # kernels named behind --no-scratch must not touch scratch memory AT ALL: in the chain kernels a scratch load sits behind a
    # vmcnt(0), which also drains the weight stream's LDS-DMA queue (round 5: a hoisted division constant did exactly that at the
    # top of every pass of the forward kernel)
    groups = {"want": [], "--no-scratch": [], "--max-inner": []}
    cur = "want"
    for a in args:
        if a in groups:
            cur = a
        else:
            groups[cur].append(a)
    want, no_scratch = groups["want"], groups["--no-scratch"]
    budget = {x.split("=")[0]: int(x.split("=")[1]) for x in groups["--max-inner"]}
    matched = {w: 0 for w in want + no_scratch + list(budget)}
    bad = 0
    for f in files:
        for name, body in kernels(open(f).read()):
            for w in matched:
                if w in name:
                    matched[w] += 1
            n_hot, vg, sg, vg_hot, sg_hot, lines, hot = analyse(body)
            if vg == 0 and sg == 0:
                continue
            # an MFMA loop whose body is short is a tile body; a spill inside a long "loop" (the job / pass loop around everything)
            # runs once per pass
            inner = [h for h in hot if sum(1 for x in lines[h[0]:h[1]] if x.startswith("v_mfma")) > 0]
            innermost = [h for h in inner if not any(o != h and h[0] <= o[0] and o[1] <= h[1] for o in inner)]
            v_in = [x for x in vg_hot if any(a <= x[0] <= b for a, b in innermost)]
            s_in = [x for x in sg_hot if any(a <= x[0] <= b for a, b in innermost)]
            print(f"{name[:60]:60s} MFMA loops {n_hot:2d} (innermost {len(innermost)})  VGPR spill ops {vg:3d} (in an innermost MFMA loop: {len(v_in)})  "
                  f"SGPR spill ops {sg:3d} (in an innermost MFMA loop: {len(s_in)})")
            if vg and any(w in name for w in no_scratch):
                bad += 1
                print("    scratch memory in a kernel that must not use it")
            if want and any(w in name for w in want) and (v_in or s_in):
                bad += 1
                for i, l, span in (v_in + s_in)[:10]:
                    print("    line", i, l[:100])
            for w, n in budget.items():
                if w in name and len(v_in) + len(s_in) > n:
                    bad += 1
                    print(f"    {len(v_in) + len(s_in)} spill operations inside an innermost MFMA loop: over the stated budget of {n}")
    for w, n in matched.items():
        if n == 0:
            bad += 1
            print(f"no kernel of {[f.split('/')[-1] for f in files]} matches '{w}': the gate would pass on nothing")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
