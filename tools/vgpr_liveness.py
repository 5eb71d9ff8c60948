#!/usr/bin/env python3
"""Physical-VGPR liveness of one kernel in a hipcc -S listing: where is the register pressure, and what is live there?

usage: tools/vgpr_liveness.py build/kernels.s <kernel-name-substring> [top]

A plain backward data-flow over the listing's basic blocks (labels, s_branch / s_cbranch_*, s_endpgm).  A write under a
partial EXEC mask is taken as a definition (it does not end the old value's life, strictly), so the counts are a lower
bound of what the allocator had to keep apart.  Prints the `top` program points of highest pressure with the source
line of the listing, and for the highest one the live registers grouped by where they were last written."""
import re
import sys

path, name = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if name in l and re.match(r"^[\w.$]+:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))

NODEF = ("global_store", "ds_write", "buffer_store", "scratch_store", "flat_store", "s_", "v_cmp", "v_readlane",
         "v_readfirstlane", "global_atomic", "ds_add_u32", "ds_add_u64", "v_nop")
DEFUSE = ("v_fmac", "v_mac", "v_writelane", "v_dot2c", "v_pk_fmac")


def vregs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"(?<![\w\[])v(\d+)\b", tok):
        out.append(int(m.group(1)))
    return out


insts = []   # (line_no, opcode, defs, uses)
labels = {}
local = {}
for i in range(start + 1, end):
    l = lines[i].split(";")[0].rstrip()
    s = l.strip()
    if not s or s.startswith("."):
        if s.endswith(":") and not s.startswith(".amd") and not s.startswith(".set"):
            labels[s[:-1]] = len(insts)
        if not s.endswith(":"):
            continue
        continue
    if re.match(r"^[\w.$]+:$", s):
        if s[:-1].isdigit():   # a local label of an inline-asm statement: may be defined more than once
            local.setdefault(s[:-1], []).append(len(insts))
        else:
            labels[s[:-1]] = len(insts)
        continue
    if s.startswith(";;") or s.startswith("//"):
        continue
    parts = s.split(None, 1)
    op = parts[0]
    ops = parts[1].split(",") if len(parts) > 1 else []
    defs, uses = [], []
    if ops:
        if op.startswith(NODEF) and not op.endswith("_rtn") and "_rtn_" not in op:
            for o in ops:
                uses += vregs(o)
            if op.startswith("global_atomic") and "sc0" in s:   # returning form
                defs += vregs(ops[0])
        else:
            defs += vregs(ops[0])
            for o in ops[1:]:
                uses += vregs(o)
            if op.startswith(DEFUSE):
                uses += vregs(ops[0])
    insts.append((i + 1, op, ops, set(defs), set(uses)))

n = len(insts)


def target(t, k):
    t = t.strip()
    if t in labels:
        return labels[t]
    m = re.match(r"^(\d+)([fb])$", t)
    if m:
        defs_ = local.get(m.group(1), [])
        if m.group(2) == "f":
            c = [x for x in defs_ if x > k]
            return min(c) if c else None
        c = [x for x in defs_ if x <= k]
        return max(c) if c else None
    return None


succ = [[] for _ in range(n)]
for k, (ln, op, ops, d, u) in enumerate(insts):
    if op == "s_endpgm":
        continue
    if op == "s_branch":
        t_ = target(ops[0], k)
        if t_ is not None:
            succ[k].append(t_)
        continue
    if op.startswith("s_cbranch"):
        t_ = target(ops[-1], k)
        if t_ is not None:
            succ[k].append(t_)
    if op == "s_setpc_b64":
        continue
    if k + 1 < n:
        succ[k].append(k + 1)

live_in = [set() for _ in range(n)]
changed = True
while changed:
    changed = False
    for k in range(n - 1, -1, -1):
        out = set()
        for s_ in succ[k]:
            out |= live_in[s_]
        new = (out - insts[k][3]) | insts[k][4]
        if new != live_in[k]:
            live_in[k] = new
            changed = True

order = sorted(range(n), key=lambda k: -len(live_in[k]))
seen_lines = []
print("instructions: %d; max live VGPRs: %d" % (n, len(live_in[order[0]])))
for k in order:
    if any(abs(k - j) < 40 for j in seen_lines):
        continue
    seen_lines.append(k)
    ln, op, ops, d, u = insts[k]
    print("line %d: %d live   %s %s" % (ln, len(live_in[k]), op, ",".join(ops)[:70]))
    if len(seen_lines) >= top:
        break
k = order[0]
print("live at the top point:", sorted(live_in[k]))

# where is each register that is live at the top point used next?  (breadth-first over the successors)
if len(sys.argv) > 4:
    import collections
    nxt = {}
    for r in sorted(live_in[k]):
        seen = {k}
        dq = collections.deque([k])
        while dq:
            x = dq.popleft()
            if r in insts[x][4]:
                nxt[r] = x
                break
            if r in insts[x][3] and x != k:
                continue
            for y in succ[x]:
                if y not in seen:
                    seen.add(y)
                    dq.append(y)
    by_line = collections.defaultdict(list)
    for r, x in nxt.items():
        by_line[x].append(r)
    for x in sorted(by_line):
        ln, op, ops, d, u = insts[x]
        print("  next use at line %d (%s %s): v%s" % (ln, op, ",".join(ops)[:50], by_line[x]))
