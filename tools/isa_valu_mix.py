"""Instruction mix of the correlation producers from the DISASSEMBLY of libbogp.so ("print DP-VALU instructions per pair").

For every instantiation of k_corr_chunk (kernel A) and k_corr_mfma (kernel A') the kernel is cut into basic blocks; the block that holds the radial
profile is the one with the `global_store_dwordx2` of r (8 stores = 8 pairs a thread in kernel A, 16 in kernel A'), the distance loop of kernel A is the
block with the most `v_add_f64` (two dimensions x 8 pairs a trip).  Printed per pair: FP64-rate VALU instructions (v_*_f64 except the transcendental
v_rsq_f64 / v_rcp_f64, which are counted apart), 32-bit VALU, matrix instructions.  Needs only hipcc's llvm-objdump: runs in the build container.
usage: python tools/isa_valu_mix.py [path/to/libbogp.so] [d]      (d: dimensions for the per-pair distance cost of kernel A, default 20)"""
import os
import re
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from support import isa_lint  # noqa: E402

KNAME = {0: "SE", 1: "Matern-1/2", 2: "Matern-3/2", 3: "Matern-5/2", 4: "abs-exp", 5: "cubic", 6: "gen-exp", 7: "Matern-nu"}


def kernels(text):
    cur, out = None, {}
    for line in text.splitlines():
        m = isa_lint._FUNC.match(line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        m = isa_lint._INSN.match(line)
        if m and cur:
            out[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def basic_blocks(insns):
    addr = {a for a, _, _ in insns}
    cuts = set()
    for i, (a, op, args) in enumerate(insns):
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm":
            if i + 1 < len(insns):
                cuts.add(insns[i + 1][0])
            m = re.search(r"<[^>]+\+0x([0-9a-fA-F]+)>", args)
            if m:
                pass  # (targets are printed relative to the function: resolved below through the raw offset when present)
    # llvm-objdump prints branch targets as a relative count of dwords: recompute them
    for i, (a, op, args) in enumerate(insns):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.match(r"(-?\d+)", args.strip())
            if m:
                off = int(m.group(1))
                if off >= 32768:
                    off -= 65536
                t = a + 4 + 4 * off
                if t in addr:
                    cuts.add(t)
    blocks, cur = [], []
    for ins in insns:
        if ins[0] in cuts and cur:
            blocks.append(cur)
            cur = []
        cur.append(ins)
    if cur:
        blocks.append(cur)
    return blocks


def classify(block):
    c = Counter()
    for _, op, _ in block:
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op in ("v_rsq_f64_e32", "v_rcp_f64_e32", "v_sqrt_f64_e32"):
            c["trans_f64"] += 1
        elif op.startswith("v_") and ("_f64" in op or op.startswith("v_ldexp_f64") or op.startswith("v_cvt_i32_f64") or "b64" in op and "mov" not in op):
            c["dp_valu"] += 1
        elif op.startswith("v_"):
            c["valu32"] += 1
        elif op.startswith("global_store"):
            c["stores"] += 1
        elif op.startswith("global_load") or op.startswith("ds_read") or op.startswith("s_load"):
            c["loads"] += 1
        if op in ("v_add_f64", "v_add_f64_e32", "v_add_f64_e64"):
            c["add_f64"] += 1
    return c


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else os.path.join(ROOT, "bayesian-optimization_amd", "libbogp.so")
    d = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 20
    print("# %s, distance cost of kernel A priced at d = %d" % (os.path.relpath(lib, ROOT), d))
    print("%-34s %9s %10s %8s %7s %6s   %s" % ("kernel", "DP VALU", "f64 trans", "VALU32", "MFMA", "pairs", "per pair: DP VALU (+ distance) | trans | VALU32 | MFMA (FMA-equivalents: 16 per 16x16x4 over 256 pairs)"))
    for text in isa_lint.disassemble_library(lib):
        for name, insns in sorted(kernels(text).items()):
            m = re.search(r"k_corr_(chunk|mfma)ILi(\d+)E(?:Li(\d+)E)?", name)
            if not m or (m.group(3) not in (None, "0")) or int(m.group(2)) > 4:  # (cubic / gen-exp / Matern-nu: the profile is not where their time is)
                continue
            kind, kid = m.group(1), int(m.group(2))
            blocks = [classify(b) for b in basic_blocks(insns)]
            prof = max(blocks, key=lambda c: (c["stores"], c["dp_valu"]))
            pairs = prof["stores"]
            if not pairs:
                continue
            label = "k_corr_%s<%s>" % (kind, KNAME.get(kid, kid))
            if kind == "chunk":
                dist = max(blocks, key=lambda c: c["add_f64"] if c["stores"] == 0 else -1)
                per_k = dist["dp_valu"] / 16.0 if dist["add_f64"] >= 16 else dist["dp_valu"] / 8.0  # the loop is unrolled by two dimensions
                extra = " + %.1f distance (%.2f a dimension x %d)" % (per_k * d, per_k, d)
                mf = 0.0
            else:
                extra = ""
                mf = float(((d + 3) // 4) * 4)  # ceil(d / 4) k-steps x 4 tiles x 2048 flop over 1024 pairs = d (padded to 4) FMA-equivalents a pair
            print("%-34s %9d %10d %8d %7d %6d   %.1f%s | %.2f | %.1f | %.1f" % (label, prof["dp_valu"], prof["trans_f64"], prof["valu32"], prof["mfma"], pairs, prof["dp_valu"] / pairs, extra,
                                                                             prof["trans_f64"] / pairs, prof["valu32"] / pairs, mf))


if __name__ == "__main__":
    main()
