"""Count instructions per s_barrier-delimited region of a kernel in a hipcc -S listing."""
import collections
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and ': ' in l][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
seg, cur = [], []
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    cur.append(t)
    if t.startswith('s_barrier'):
        seg.append(cur)
        cur = []
seg.append(cur)
for i, s in enumerate(seg):
    c = collections.Counter()
    for t in s:
        op = t.split()[0]
        if op.startswith('v_'):
            c['valu'] += 1
            if 'dpp' in t:
                c['dpp'] += 1
            if op in ('v_rcp_f32', 'v_div_scale_f32', 'v_div_fmas_f32', 'v_div_fixup_f32'):
                c['div'] += 1
            if op.startswith('v_pk_'):
                c['pk'] += 1
        elif op.startswith(('ds_bpermute', 'ds_swizzle', 'ds_permute')):
            c['ds_perm'] += 1
        elif op.startswith('ds_'):
            c['ds'] += 1
        elif op.startswith(('global_', 'buffer_')):
            c['vmem'] += 1
        elif op.startswith('scratch_'):
            c['scratch'] += 1
        elif op.startswith('s_waitcnt'):
            c['wait'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
    print(i, len(s), dict(c))
