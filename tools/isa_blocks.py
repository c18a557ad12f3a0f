"""Opcode table of one kernel from an ISA dump (hipcc --cuda-device-only -S file.hip -o file.s): per basic block the
instruction count by class - MFMA, transcendental, conversions, other VALU, LDS, VMEM / LDS-DMA, SALU - so that the
steady-state path of a loop can be read off and priced ("dump the ISA, count the softmax phase by opcode").

  python tools/isa_blocks.py file.s KERNEL_SUBSTRING [min_block_size]"""
import collections, re, sys

f, kern = sys.argv[1], sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(f).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kern) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_cvt_pk_bf16"): return "cvt_pk_bf16"
    if op.startswith("v_pk_"): return "v_pk"
    if op.startswith(("v_max", "v_min")): return "max/min"
    if op.startswith(("v_mov", "v_accvgpr")): return "mov"
    if op.startswith(("v_cmp", "v_cndmask")): return "cmp/select"
    if op.startswith("v_permlane") or "dpp" in op: return "lane-xchg"
    if op.startswith("v_"): return "valu-other"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    return "salu"


order = ["mfma", "trans", "cvt_pk_bf16", "v_pk", "max/min", "mov", "cmp/select", "lane-xchg", "valu-other", "lds", "vmem", "salu",
         "waitcnt", "nop", "barrier", "branch"]
blocks, cur = [], ["(entry)", collections.Counter(), []]
blocks.append(cur)
for l in lines[start + 1:end]:
    l = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), collections.Counter(), []]
        blocks.append(cur)
        continue
    if not l or l.startswith((";", ".")):
        continue
    op = l.split()[0]
    cur[1][cls(op)] += 1
    if op.startswith(("s_cbranch", "s_branch")):
        cur[2].append(l.split(";")[0].split()[-1])
vg = next((l.strip() for l in lines[end:end + 120] if "NumVgprs" in l), "")
print(f"kernel *{kern}*: {sum(sum(b[1].values()) for b in blocks)} instructions, {vg}")
print(f"{'block':12s} {'n':>4s}  " + " ".join(f"{o:>11s}" for o in order) + "  -> branches to")
for name, c, br in blocks:
    n = sum(c.values())
    if n < min_n:
        continue
    print(f"{name:12s} {n:4d}  " + " ".join(f"{c.get(o, 0):11d}" for o in order) + "  -> " + ",".join(br))
