"""Compressed view of a kernel's instruction stream from a hipcc -S dump (development aid).
Usage: python tools/isa_summary.py file.s <substring of kernel symbol> [max_lines]"""
import re
import sys

s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 200
start = next(i for i, l in enumerate(s) if l.startswith('_Z') and key in l.split(':')[0] and l.rstrip().split(';')[0].rstrip().endswith(':'))
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
seq = []
for l in s[start + 1:end]:
    l = l.strip()
    if not l or l.startswith(';'):
        continue
    op = l.split()[0]
    if op.startswith(('v_mfma', 's_waitcnt', 'global_load', 's_barrier', 's_cbranch', 'ds_read', 'ds_write', '.LBB', 'global_store', 'buffer_', 'scratch_')):
        k = l.split(';')[0].strip() if op.startswith(('s_waitcnt', '.LBB', 's_cbranch')) else op
        if seq and seq[-1][0] == k:
            seq[-1][1] += 1
        else:
            seq.append([k, 1])
print(end - start, "lines")
for k, c in seq[:mx]:
    print(c, k)
