"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (VALU / MFMA / global loads / scratch / vmcnt waits).
   python tools/isa_blocks.py <file.s> <substring of the mangled kernel name>"""
import re, sys
s = open(sys.argv[1]).read()
for f in re.split(r'\n(?=_Z[\w]+:)', s):
    name = f.split(':')[0]
    if sys.argv[2] not in name:
        continue
    for b in re.split(r'\n(?=\.LBB\d+_\d+:)', f):
        lines = [l.strip() for l in b.split('\n')]
        ins = [l for l in lines if l and not l.startswith(('.', ';', '_Z')) and not l.endswith(':')]
        nv = sum(1 for i in ins if i.startswith('v_') and not i.startswith('v_mfma'))
        nm = sum(1 for i in ins if i.startswith('v_mfma'))
        ng = sum(1 for i in ins if i.startswith('global_load'))
        nsc = sum(1 for i in ins if i.startswith('scratch_'))
        wc = [i.split('vmcnt')[1][:4] for i in ins if i.startswith('s_waitcnt') and 'vmcnt' in i]
        if nv > 20 or ng > 3:
            print(lines[0][:14], 'valu', nv, 'mfma', nm, 'gload', ng, 'scratch', nsc, 'total', len(ins), 'vmwaits', wc)
