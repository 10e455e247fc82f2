"""Opcode histogram of a kernel's biggest loop from `hipcc -S --cuda-device-only` output.

    python tools/isa_count.py file.s <mangled-kernel-name-substring> [--all | --loop .LBBn_m]

Prints, for the loop with the most instructions (or the whole kernel with --all): vector
instructions by opcode and by number of VGPR source operands, LDS / memory / scalar counts.
Used for the before / after instruction counts quoted in DESIGN.md section 4.2.
"""
import collections
import re
import sys


def kernel_body(lines, name):
    start = next(i for i, l in enumerate(lines) if re.match(r"^\S*" + re.escape(name) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def loops(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i, m.group(1)))
    return out


def biggest_loop(body, label=None):
    """The loop with the most vector instructions that is not just a wrapper around another
    loop holding 60 % of them (a kernel's outer bin / chunk loops), or the loop at `label`."""
    ls = loops(body)
    if label:
        a, b, _ = max((l for l in ls if l[2] == label), key=lambda l: l[1] - l[0])
        return body[a:b + 1]
    if not ls:
        return body
    valu = {l: classify(body[l[0]:l[1] + 1])[0]["valu"] for l in ls}
    for l in sorted(ls, key=lambda l: -valu[l]):
        inner = [m for m in ls if m != l and l[0] <= m[0] and m[1] <= l[1] and (m[0], m[1]) != (l[0], l[1])]
        if not any(valu[m] >= 0.6 * valu[l] for m in inner):
            return body[l[0]:l[1] + 1]
    return body


def classify(body):
    ops = collections.Counter()
    vsrc = collections.Counter()
    cls = collections.Counter()
    for l in body:
        s = l.strip()
        if not l.startswith("\t") or not s or s[0] in ";.":
            continue
        s = s.split(";")[0].strip()
        op = s.split()[0]
        args = s[len(op):]
        if op.startswith("v_"):
            cls["valu"] += 1
            ops[op] += 1
            parts = [a.strip() for a in args.split(",")]
            n_src = sum(1 for a in parts[1:] if re.match(r"^-?\|?v(\d+|\[)", a))
            if op.startswith(("v_fmac", "v_mac")):
                n_src += 1  # the destination is read too
            vsrc[n_src] += 1
        elif op.startswith("ds_"):
            cls["lds"] += 1
            ops[op] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cls["vmem"] += 1
            ops[op] += 1
        elif op.startswith("s_"):
            cls["salu"] += 1
            if op in ("s_barrier", "s_waitcnt", "s_nop", "s_setprio"):
                ops[op] += 1
    return cls, ops, vsrc


def main():
    lines = open(sys.argv[1]).read().split("\n")
    body = kernel_body(lines, sys.argv[2])
    if "--all" not in sys.argv:
        label = sys.argv[sys.argv.index("--loop") + 1] if "--loop" in sys.argv else None
        body = biggest_loop(body, label)
    cls, ops, vsrc = classify(body)
    print("classes:", dict(cls))
    print("VALU by VGPR sources read:", dict(sorted(vsrc.items())))
    for op, n in sorted(ops.items(), key=lambda kv: -kv[1]):
        print(f"  {op:28s} {n}")


if __name__ == "__main__":
    main()
