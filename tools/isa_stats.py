"""Instruction mix per basic block of one kernel in a hipcc -S dump (tools: kernel tuning)."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2]
start = [m for m in re.finditer(r'^(_Z\S*' + pat + r'\S*):\s*(;.*)?$', s, re.M)][0]
end = s.index('s_endpgm', start.end())
body = s[start.end():end]
lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((';', '.p2align', '.loc', '.cfi'))]
labels = [i for i, l in enumerate(lines) if re.match(r'^\.?LBB\S*:', l)]
def cat(l):
    op = l.split()[0]
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_'): return 'salu'
    return 'other'
idx = [0] + labels + [len(lines)]
for a, b in zip(idx, idx[1:]):
    c = Counter(cat(l) for l in lines[a + 1:b])
    if b - a > 12:
        print(lines[a][:30].ljust(30), b - a - 1, dict(c))
tail = s[end:end + 4000]
print(re.findall(r'; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs): (\d+)', tail))
