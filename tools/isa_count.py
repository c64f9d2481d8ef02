"""Count instruction classes in one kernel of a hipcc -S listing (whole kernel body; the scan loop dominates it).
usage: isa_count.py listing.s kernel-substring"""
import re, sys, collections
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(\S*%s\S*):[^\n]*\n(.*?)\n\.Lfunc_end' % re.escape(pat), s, re.S | re.M)
body = m.group(2)
c = collections.Counter()
ops = collections.Counter()
for line in body.split('\n'):
    line = line.strip()
    if not line or line.startswith(('.', ';')) or line.endswith(':'):
        continue
    op = line.split()[0]
    ops[op] += 1
    if op.startswith('v_pk_'): c['v_pk'] += 1
    elif op.startswith('v_mov') or op.startswith('v_accvgpr'): c['v_mov'] += 1
    elif op.startswith('v_'): c['valu_other'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith('global_') or op.startswith('buffer_'): c['vmem'] += 1
    elif op.startswith('scratch_'): c['scratch'] += 1
    elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    else: c['other'] += 1
print(m.group(1)[:70], dict(c), 'VALU total', c['v_pk'] + c['v_mov'] + c['valu_other'])
if len(sys.argv) > 3:
    for k, v in ops.most_common(25): print('   ', k, v)
