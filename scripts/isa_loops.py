"""Developer script: the loops of a kernel in a disassembly (llvm-objdump -d): every backward branch with the number of
instructions of its body, how many of them are scratch / LDS / global / DPP, and its nesting.  scripts/isa_loops.py file.s symbol"""
import re
import sys
lines = [l.rstrip('\n') for l in open(sys.argv[1])]
sym = sys.argv[2]
start = [i for i, l in enumerate(lines) if sym in l and l.endswith('>:')][0]
ins = []
for l in lines[start + 1:]:
    m = re.match(r'\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):', l)
    if m:
        t = re.search(r'<[^+>]*\+0x([0-9a-f]+)>\s*$', l)
        ins.append((int(m.group(3), 16), m.group(1), m.group(2), int(t.group(1), 16) if t else None))
    elif re.match(r'^[0-9a-f]+ <', l):
        break
base = ins[0][0]
idx = {a - base: i for i, (a, _, _, _) in enumerate(ins)}
loops = []
for i, (a, op, args, tgt) in enumerate(ins):
    if (op.startswith('s_cbranch') or op == 's_branch') and tgt is not None and tgt in idx and idx[tgt] <= i:
        loops.append((idx[tgt], i))
loops.sort()
def cls(op):
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('flat_'): return 'vmem'
    if 'dpp' in op: return 'dpp'
    if op.startswith('v_') and 'f64' in op: return 'fp64'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    return 'other'
print(len(ins), 'instructions;', len(loops), 'loops')
for (s, e) in loops:
    depth = sum(1 for (s2, e2) in loops if s2 <= s and e2 >= e and (s2, e2) != (s, e))
    c = {}
    for k in range(s, e + 1):
        c[cls(ins[k][1])] = c.get(cls(ins[k][1]), 0) + 1
    print('  ' * depth + 'loop @%d..%d: %d instr' % (s, e, e - s + 1), ' '.join('%s %d' % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])))
