"""Instruction mix of a kernel's main loop from hipcc -S output: the loop = the widest backward branch that contains an s_barrier.
usage: python tools/isa_loop.py file.s <substring of the mangled kernel name>"""
import collections
import re
import sys

def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    m = re.search(r'^(_Z\w*%s\w*):' % re.escape(key), s, re.M)
    start = m.start()
    end = s.index('.Lfunc_end', start)
    lines = [l.split(';')[0].strip() for l in s[start:end].split('\n')]
    lines = [l for l in lines if l and not l.startswith(';') and not (l.startswith('.') and not l.endswith(':'))]
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(':')}
    loops = []
    for i, l in enumerate(lines):
        if l.startswith('s_cbranch') or l.startswith('s_branch'):
            t = labels.get(l.split()[-1])
            if t is not None and t < i and any('s_barrier' in x for x in lines[t:i]) and any('v_mfma' in x for x in lines[t:i]):
                loops.append((t, i))
    which = sys.argv[3] if len(sys.argv) > 3 else 'widest'          # widest | all | <index by size, 0 = smallest>
    loops.sort(key=lambda r: r[1] - r[0])
    if which == 'all':
        for n, (t, i) in enumerate(loops):
            print(n, t, i, i - t)
        return
    t, i = loops[-1] if which == 'widest' else loops[int(which)]
    body = [l for l in lines[t:i + 1] if not l.endswith(':')]
    c = collections.Counter(l.split()[0] for l in body)
    valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
    print(f"{m.group(1)}: loop of {len(body)} instructions; VALU {valu}, MFMA {sum(v for k, v in c.items() if 'mfma' in k)}, "
          f"SALU {sum(v for k, v in c.items() if k.startswith('s_'))}, DS {sum(v for k, v in c.items() if k.startswith('ds_'))}")
    for k, v in c.most_common(40):
        print(f"  {v:4d} {k}")

if __name__ == "__main__":
    main()
