#!/usr/bin/env python3
"""Static check of the hand-managed memory operations in the band kernels' generated code.

    tools/check_isa.py FILE.s [FILE.s ...]      (device assembly: hipcc -S, or -save-temps of the library build)

The band kernels (csrc/band_kernels.hpp, csrc/band32_kernels.hpp) issue their row requests from inline
assembly and wait for them with hand-placed `s_waitcnt vmcnt(N)`.  The compiler neither sees that a load is
in flight nor looks inside the assembly for hazards, so two properties hold only as long as its register
allocation and scheduling happen to respect them.  This script checks them in the code that was actually
generated, for every kernel whose name contains `k_band`:

1. every inline-assembly `buffer_load` is immediately preceded by its `s_nop 4`: an SGPR of the resource (or
   M0, for the LDS-DMA form) written by a VALU instruction (`v_readlane` restoring a spilled SGPR,
   `v_readfirstlane`) must not be read by a VMEM instruction within 5 wait states, and the hazard recogniser
   skips inline assembly.  With the nop in front, nothing earlier can be closer than that.
2. no instruction reads or writes a VGPR / AGPR that is the destination of an INLINE-ASSEMBLY load still in flight: on every
   path from a load to a use of its destination there is an `s_waitcnt vmcnt(N)` with N small enough to
   cover it (vmcnt counts every vector memory operation in issue order: a load with k younger ones is complete
   after `vmcnt(N)`, N <= k).  A copy or a spill of such a register between request and wait — legal for the
   compiler, which believes the asm statement defined it — would silently read stale data.  (Loads that target
   the LDS have no register destination: only property 1 applies to them.)

Exit status 0 when both hold everywhere, 1 with one line per violation otherwise.
"""
import re
import sys

VMEM = re.compile(r"^(buffer|global|flat|scratch)_(load|store|atomic)")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
BRANCH = re.compile(r"^s_c?branch\S*\s+(\S+)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add(m.group(1) + m.group(2))
        else:
            for k in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add(m.group(3) + str(k))
    return out


class Ins:
    __slots__ = ("text", "op", "in_asm", "line", "dests", "uses", "is_vmem", "vmcnt", "target", "is_uncond", "ends")

    def __init__(self, text, in_asm, line):
        self.text, self.in_asm, self.line = text, in_asm, line
        self.op = text.split()[0]
        self.is_vmem = bool(VMEM.match(self.op))
        self.dests, self.uses = set(), regs_of(text)
        if self.is_vmem and in_asm and "_load" in self.op and not re.search(r"\blds\b", text):  # (the compiler's own loads: its job)
            first = text.split(None, 1)[1].split(",")[0]
            self.dests = regs_of(first)
        self.vmcnt = None
        if self.op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", text)
            if m:
                self.vmcnt = int(m.group(1))
            elif re.fullmatch(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)", text.strip()):  # raw immediate: treat as a full wait
                self.vmcnt = 0
        m = BRANCH.match(text)
        self.target = m.group(1) if m else None
        self.is_uncond = self.op == "s_branch"
        self.ends = self.op in ("s_endpgm", "s_trap")


def kernels(path):
    """{name: ([Ins], {label: index})} for every function of the file"""
    out, name, body, labels, in_asm = {}, None, None, None, False
    for ln, raw in enumerate(open(path, errors="replace"), 1):
        line = raw.split(";")[0].rstrip() if ";;#ASM" not in raw else raw.rstrip()
        if ";;#ASMSTART" in raw:
            in_asm = True
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            name, body, labels = m.group(1), [], {}
            out[name] = (body, labels)
            continue
        if name is None:
            continue
        m = re.match(r"^(\.L\w+):", line)
        if m:
            labels[m.group(1)] = len(body)
            continue
        s = line.strip()
        if not s or s.startswith(".") or s.startswith("//"):
            if s.startswith(".end_amdhsa_kernel") or s.startswith(".section"):
                name = None
            continue
        body.append(Ins(s, in_asm, ln))
        if body[-1].ends and False:
            name = None
    return out


def check_kernel(name, body, labels):
    problems = []
    n = len(body)
    # 1. s_nop 4 in front of every inline-assembly buffer load
    n_inline = 0
    for i, ins in enumerate(body):
        if ins.in_asm and ins.op.startswith("buffer_load"):
            n_inline += 1
            if i == 0 or not re.fullmatch(r"s_nop\s+4", body[i - 1].text.strip()) or not body[i - 1].in_asm:
                problems.append(f"{name}: line {ins.line}: `{ins.text}` is not preceded by its `s_nop 4`")
    # 2. destinations of loads in flight: forward dataflow, state = {reg: younger VMEM operations issued so far}
    succ = [[] for _ in range(n)]
    for i, ins in enumerate(body):
        if ins.ends:
            continue
        if ins.target is not None:
            if ins.target not in labels:
                problems.append(f"{name}: line {ins.line}: branch to unknown label {ins.target}")
            else:
                succ[i].append(labels[ins.target])
            if not ins.is_uncond and i + 1 < n:
                succ[i].append(i + 1)
        elif i + 1 < n:
            succ[i].append(i + 1)
    state_in = [None] * n
    state_in[0] = {}
    work = [0]
    reported = set()
    while work:
        i = work.pop()
        st = dict(state_in[i])
        ins = body[i]
        # (a load may target a register whose previous load is still in flight: loads complete in issue order)
        touched = ((ins.uses - ins.dests) if ins.dests else ins.uses) & st.keys()
        if touched and (ins.line, tuple(sorted(touched))) not in reported:
            reported.add((ins.line, tuple(sorted(touched))))
            problems.append(f"{name}: line {ins.line}: `{ins.text}` touches {sorted(touched)} while the load that "
                            f"targets it may still be in flight (no covering s_waitcnt vmcnt on some path)")
        if ins.is_vmem:
            st = {r: a + 1 for r, a in st.items()}
            for r in ins.dests:
                st[r] = 0
        elif ins.vmcnt is not None:
            st = {r: a for r, a in st.items() if a < ins.vmcnt}
        else:
            for r in touched:  # reported once; do not cascade
                st.pop(r, None)
        for j in succ[i]:
            old = state_in[j]
            if old is None:
                state_in[j] = dict(st)
                work.append(j)
            else:
                changed = False
                for r, a in st.items():
                    if r not in old or a < old[r]:
                        old[r] = a
                        changed = True
                if changed:
                    work.append(j)
    return n_inline, problems


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    bad, seen = 0, 0
    for path in argv[1:]:
        for name, (body, labels) in kernels(path).items():
            if "k_band" not in name or not body:
                continue
            n_inline, problems = check_kernel(name, body, labels)
            if n_inline == 0 and "gather" not in name:
                problems.append(f"{name}: no inline-assembly buffer load found (wrong file, or the kernel changed: update this check)")
            seen += 1
            short = re.sub(r"^_ZN2ta\d+", "", name)[:40]
            print(f"check_isa: {short:42s} {len(body):6d} instructions, {n_inline:3d} inline loads: "
                  + ("ok" if not problems else f"{len(problems)} PROBLEM(S)"))
            for p in problems[:20]:
                print("  " + p)
            bad += len(problems)
    if seen == 0:
        print("check_isa: no k_band kernel in the given files")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
