"""Static instruction census of one function of a gfx950 assembly listing (hipcc -S --cuda-device-only): per loop (a label that is the
target of a backward branch ... that branch) the instruction counts by class.  CPU only.
python tools/isa_census.py /tmp/scp_api.s _ZN3scp14ipm2_ph_factorINS_13RocketLandingELi2EEEvPKdPdidS4_ [min_loop_size]"""
import re, sys, collections
path, fn = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(fn + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
def cls(op, rest):
    if op.startswith("v_mfma"): return "mfma"
    if op.endswith("_f64") or "_f64_" in op:
        if op.startswith(("v_fma", "v_mul", "v_add", "v_fmac")): return "fp64_arith"
        if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_div", "v_trig", "v_ldexp", "v_frexp")): return "fp64_special"
        if op.startswith(("v_cmp", "v_max", "v_min")): return "fp64_cmp/minmax"
        return "fp64_other"
    if "dpp" in rest or op.endswith("_dpp"): return "dpp"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): return "lane<->sgpr"
    if op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"): return "ds_permute"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "lds_write"
    if op.startswith("global_load") or op.startswith("flat_load") or op.startswith("buffer_load"): return "vmem_load"
    if op.startswith("global_store") or op.startswith("flat_store") or op.startswith("buffer_store"): return "vmem_store"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_cndmask"): return "v_cndmask"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "v_mov"
    if op.startswith("v_cmp"): return "v_cmp(int)"
    if op.startswith("v_"): return "valu_int/other"
    return "other"
ins = []          # (index, label or None, op, rest)
labels = {}
for l in body:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        labels[m.group(1)] = len(ins); continue
    m = re.match(r"^\t([a-z_0-9]+)\s*(.*)$", l)
    if m and not m.group(1).startswith(("p2align", "amdhsa", "section", "globl", "type", "size", "text")) and not l.startswith("\t."):
        ins.append((m.group(1), m.group(2).split(";")[0]))
print("function: %d instructions" % len(ins))
tot = collections.Counter(cls(o, r) for o, r in ins)
print("  whole function:", dict(tot.most_common()))
loops = []
for i, (o, r) in enumerate(ins):
    if o.startswith("s_cbranch") or o == "s_branch":
        t = r.strip()
        if t in labels and labels[t] <= i and i - labels[t] >= minsz:
            loops.append((labels[t], i, t))
loops.sort(key=lambda x: (x[0], -x[1]))
for a, b, t in loops:
    c = collections.Counter(cls(o, r) for o, r in ins[a:b + 1])
    n = b - a + 1
    print("loop %s: %d instructions [%d..%d]" % (t, n, a, b))
    print("   " + ", ".join("%s %d (%.0f%%)" % (k, v, 100.0 * v / n) for k, v in c.most_common()))
