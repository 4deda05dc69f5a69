"""Instruction histogram of one kernel in a hipcc --save-temps .s file.

  python tools/isa_hist.py FILE.s SUBSTRING [--loop | --inner] [--dump OUT.s]

SUBSTRING selects the kernel by (mangled) name; --loop restricts the count to
the largest loop (label .. backward branch to it), --inner to the largest
loop that contains no other loop.
"""
import collections
import re
import sys


def kernel_body(lines, sub):
  starts = [i for i, l in enumerate(lines)
            if re.match(r'^_Z\S*:', l) and sub in l]
  if not starts:
    raise SystemExit(f'no kernel matching {sub!r}')
  s = starts[0]
  e = next(i for i in range(s, len(lines))
           if lines[i].strip().startswith('.end_amdhsa_kernel')
           or lines[i].strip().startswith('.Lfunc_end'))
  return lines[s:e]


def loops(body):
  labels = {l.split(':')[0]: i for i, l in enumerate(body)
            if re.match(r'^\.LBB\S*:', l)}
  out = []
  for i, l in enumerate(body):
    m = re.search(r's_cbranch\S*\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)', l)
    if m:
      tgt = labels.get(m.group(1) or m.group(2))
      if tgt is not None and tgt < i:
        out.append((tgt, i + 1))
  return out


def biggest_loop(body, inner=False):
  ls = loops(body)
  if inner:  # loops that contain no other loop
    ls = [a for a in ls
          if not any(b != a and a[0] <= b[0] and b[1] <= a[1] for b in ls)]
  if not ls:
    return body
  a = max(ls, key=lambda t: t[1] - t[0])
  return body[a[0]:a[1]]


def main():
  path, sub = sys.argv[1], sys.argv[2]
  lines = open(path).read().split('\n')
  body = kernel_body(lines, sub)
  if '--dump' in sys.argv:
    open(sys.argv[sys.argv.index('--dump') + 1], 'w').write('\n'.join(body))
  if '--loop' in sys.argv:
    body = biggest_loop(body)
  if '--inner' in sys.argv:
    body = biggest_loop(body, inner=True)
  c = collections.Counter()
  for l in body:
    l = l.strip()
    if not l or l[0] in ';.' or l.endswith(':'):
      continue
    c[l.split()[0]] += 1
  valu = sum(v for k, v in c.items() if k.startswith('v_'))
  lds = sum(v for k, v in c.items() if k.startswith('ds_'))
  vmem = sum(v for k, v in c.items()
             if k.startswith(('global_', 'buffer_', 'flat_')))
  salu = sum(v for k, v in c.items() if k.startswith('s_'))
  print(f'total {sum(c.values())}  VALU {valu}  LDS {lds}  VMEM {vmem}  '
        f'SALU {salu}')
  for k, v in c.most_common(50):
    print(f'  {k:28s} {v}')


if __name__ == '__main__':
  main()
