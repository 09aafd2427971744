"""Instruction mix of the innermost loops around every s_barrier of a kernel in an AMDGPU assembly listing (diagnostic).

usage: python tools/isa_loops.py file.s [kernel-name-substring]"""
import re
import sys


def main():
    text = open(sys.argv[1]).read().split('\n')
    key = sys.argv[2] if len(sys.argv) > 2 else None
    if key:
        start = next(i for i, l in enumerate(text) if l.startswith('_Z') and key in l and ':' in l)
        end = next(i for i in range(start, len(text)) if 's_endpgm' in text[i])
        text = text[start:end + 1]
    labels = {}
    for i, l in enumerate(text):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(text):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    pats = [('mfma', r'v_mfma'), ('accread', r'v_accvgpr_read'), ('accwrite', r'v_accvgpr_write'), ('valu', r'^\s+v_(?!mfma|accvgpr)'),
            ('ds', r'^\s+ds_'), ('gload', r'global_load|flat_load'), ('gstore', r'global_store|flat_store'), ('waitcnt', r's_waitcnt'),
            ('nop', r's_nop'), ('scratch', r'scratch_')]
    bars = [i for i, l in enumerate(text) if 's_barrier' in l]
    for b in bars:
        cands = [(a, e) for a, e in loops if a <= b <= e]
        if not cands:
            print('barrier at', b, 'not inside a loop')
            continue
        a, e = min(cands, key=lambda x: x[1] - x[0])
        seg = text[a:e + 1]
        out = [f'loop lines {a}-{e}']
        for name, p in pats:
            out.append(f'{name} {sum(1 for l in seg if re.search(p, l))}')
        print('  '.join(out))


if __name__ == '__main__':
    main()
