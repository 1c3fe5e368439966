"""(part of the round-6 experiment r06_k3_fused_dpp_fma.patch -- not used by the shipped build)
Static check of a gfx950 assembly listing (hipcc -S --cuda-device-only) for the one hazard inline-assembly DPP instructions take over from
the compiler: a DPP instruction must not read a VGPR that a VALU instruction wrote less than 2 wait states earlier (an instruction = 1
wait state, `s_nop N` = N + 1).  The compiler's hazard recogniser does not look into inline assembly.  It found six real violations in the
first version of the experiment (a rematerialised constant written between two operations of a group).  CPU only; exit code 1 on a violation.
    python profiles/experiments/dpp_hazard_check.py /tmp/scp_api.s"""
import re
import sys


def regs(tok):
    """VGPR indices named by an operand like v12, v[4:5], -v[4:5], |v3|."""
    m = re.search(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def main():
    path = sys.argv[1]
    bad = dpp = 0
    hist = []          # (wait states this instruction provides, set of VGPRs it writes as a VALU instruction)
    fn = "?"
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if not t or t.startswith((";", "//", ".")) and not t.startswith(".LBB"):
            continue
        if re.match(r"^[A-Za-z_.$][\w.$]*:", t):
            if not t.startswith(".L"):
                fn = t[:-1]
            hist = []          # block boundary: the predecessors are not known here
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if op == "s_nop":
            hist.append((int(ops[0], 0) + 1, set()))
            continue
        is_dpp = "row_newbcast" in t or "_dpp" in op or "quad_perm" in t or "row_shr" in t or "row_shl" in t or "row_bcast" in t
        if is_dpp and len(ops) >= 2:
            dpp += 1
            src = regs(ops[1])
            ws = 0
            for w, wr in reversed(hist):
                if ws >= 2:
                    break
                if wr & src:
                    bad += 1
                    print("%s:%d: %s -- DPP operand written %d wait state(s) earlier (in %s)" % (path, ln, t, ws, fn))
                    break
                ws += w
        wr = set()
        if op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and ops:
            wr = regs(ops[0])
        hist.append((1, wr))
        if len(hist) > 8:
            hist = hist[-8:]
    print("%d DPP instructions checked, %d hazards" % (dpp, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
