"""Instruction mix per basic block of one kernel of a hipcc -S listing: python scratch/isa_blocks.py file.s <mangled-name-prefix> [min-instructions]"""
import re, sys, collections
L = open(sys.argv[1]).read().split('\n')
pre = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 60
start = [i for i, l in enumerate(L) if l.startswith(pre) and ':' in l.split(';')[0]][0]
end = [i for i, l in enumerate(L) if i > start and l.startswith('.Lfunc_end')][0]
cur = 'entry'; blocks = collections.OrderedDict(); blocks[cur] = []
for l in L[start + 1:end]:
    s = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        cur = m.group(1); blocks[cur] = []; continue
    if not s or s.startswith(('.', ';', '//')): continue
    blocks[cur].append(s.split()[0])
for n, ins in blocks.items():
    c = collections.Counter(ins)
    if len(ins) >= mn:
        print(n, len(ins), 'mfma', sum(v for k, v in c.items() if 'mfma' in k), 'valu', sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k), 'ds', sum(v for k, v in c.items() if k.startswith('ds_')),
              'global', sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_'))))
        print('    ', ' '.join('%s:%d' % kv for kv in c.most_common(30)))
