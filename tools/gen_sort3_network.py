"""Sorting networks from 2-sorters AND 3-sorters for the K3 register sort.

gfx950 has `v_min3_f32` / `v_med3_f32` / `v_max3_f32`: a 3-sorter costs 3 VALU
instructions for 3 elements where three 2-sorters (`v_min_f32` + `v_max_f32`
each) cost 6.  A 3-way odd-even merge sort (sort triples, 3-way merge the
columns recursively, a 7-instruction-per-row clean-up) therefore needs fewer
instructions per element than Batcher's 2-way network.

  python tools/gen_sort3_network.py            # report + verification
  python tools/gen_sort3_network.py --emit     # rewrite csrc/sort3_network_50.inc

Networks are lists of ops on abstract wires; an op orders its arguments
(`s2(a, b)`: a <= b afterwards; `s3(a, b, c)`: a <= b <= c).  Because the
program is fully unrolled over registers, the sorted order is a PERMUTATION of
the wires (returned as a list) -- no data movement is ever emitted.

Verification (0-1 principle): the 27-sorter exhaustively on all 2^27 0/1 inputs
(bit-sliced: min = AND, max = OR, med = majority), every merge on all sorted
0/1 inputs; plus random real inputs through the pruned 50-input program.
"""
import itertools
import os
import sys

import numpy as np


class Net:

  def __init__(self):
    self.ops = []

  def s2(self, a, b):
    self.ops.append((a, b))

  def s3(self, a, b, c):
    self.ops.append((a, b, c))


def merge2(net, A, B):
  """Batcher's odd-even merge of two sorted wire lists of any lengths."""
  if not A or not B:
    return list(A) + list(B)
  if len(A) == 1 and len(B) == 1:
    net.s2(A[0], B[0])
    return [A[0], B[0]]
  ev = merge2(net, A[0::2], B[0::2])
  od = merge2(net, A[1::2], B[1::2])
  # interleave: ev[0], then pairs (od[i], ev[i + 1]) compare-exchanged
  out = [ev[0]]
  for i in range(max(len(od), len(ev) - 1)):
    o = od[i] if i < len(od) else None
    e = ev[i + 1] if i + 1 < len(ev) else None
    if o is not None and e is not None:
      net.s2(o, e)
      out += [o, e]
    elif o is not None:
      out.append(o)
    else:
      out.append(e)
  return out


def _layer3(net, E, offset):
  """3-sorters on (E[offset + 3i], +1, +2); a trailing pair gets a 2-sorter."""
  i = offset
  while i + 2 < len(E):
    net.s3(E[i], E[i + 1], E[i + 2])
    i += 3
  if i + 1 < len(E):
    net.s2(E[i], E[i + 1])


def merge3(net, A, B, C):
  """3-way odd-even merge of three sorted wire lists of EQUAL length 3^k."""
  n = len(A)
  assert len(B) == n and len(C) == n
  if n == 1:
    net.s3(A[0], B[0], C[0])
    return [A[0], B[0], C[0]]
  assert n % 3 == 0
  D = [merge3(net, A[r::3], B[r::3], C[r::3]) for r in range(3)]
  E = [D[r][i] for i in range(n) for r in range(3)]
  # Column r of the interleaved rows has ceil((z_L - r) / 3) zeros per list L:
  # rows are (0,0,0) ... up to 3 dirty rows (0,0,1) / (0,1,1) ... (1,1,1).  The
  # cheapest clean-up found by search (7 instructions per row; none with 6 in
  # this template family; full layers of 3-sorters need three = 9): col 1 of a
  # row against col 0 of the next, col 2 against col 1 of the next, then
  # 3-sorters on (3i+1, 3i+2, 3i+3) -- verified on all sorted 0/1 inputs below.
  for i in range(n - 1):
    net.s2(E[3 * i + 1], E[3 * i + 3])
  for i in range(n - 1):
    net.s2(E[3 * i + 2], E[3 * i + 4])
  _layer3(net, E, 1)
  return E


def sort3k(net, wires):
  """Sorts 3^k wires with the 3-way merge sort."""
  n = len(wires)
  if n == 1:
    return list(wires)
  assert n % 3 == 0
  third = n // 3
  parts = [sort3k(net, wires[i * third:(i + 1) * third]) for i in range(3)]
  return merge3(net, *parts)


def sort_general(net, wires):
  """Any length: 3^k blocks by the 3-way sort, the rest by 2-way merges."""
  n = len(wires)
  if n <= 1:
    return list(wires)
  if n == 2:
    net.s2(wires[0], wires[1])
    return list(wires)
  p = 1
  while p * 3 <= n:
    p *= 3
  if p == n:
    return sort3k(net, wires)
  return merge2(net, sort_general(net, wires[:p]), sort_general(net, wires[p:]))


def batcher_sort(net, wires):
  if len(wires) <= 1:
    return list(wires)
  h = (len(wires) + 1) // 2
  return merge2(net, batcher_sort(net, wires[:h]), batcher_sort(net, wires[h:]))


# ---------------------------------------------------------------------------
def prune(ops, order, n_wires, n_real):
  """Wires >= n_real hold +inf: ops touching them shrink (3-sorter -> 2-sorter
  -> nothing) and the wire NAMES are renamed instead of moving data.  Returns
  (ops on real registers, order of the first n_real ranks)."""
  name = list(range(n_wires))          # wire -> register id (or None = +inf)
  for w in range(n_real, n_wires):
    name[w] = None
  out = []
  for op in ops:
    regs = [name[w] for w in op]
    real = [r for r in regs if r is not None]
    if len(real) >= 2:
      out.append(tuple(real))
    # after the op the smallest values sit on the first wires: reals first
    for w, r in zip(op, real + [None] * (len(op) - len(real))):
      name[w] = r
  final = [name[w] for w in order]
  assert all(r is not None for r in final[:n_real]), 'inf inside the real ranks'
  assert all(r is None for r in final[n_real:])
  return out, final[:n_real]


def cost(ops):
  return sum(len(op) for op in ops)  # 2 or 3 instructions per op


def run(ops, x):
  """Applies ops to the columns of x (wires along axis 0)."""
  x = x.copy()
  for op in ops:
    if len(op) == 2:
      a, b = op
      lo, hi = np.minimum(x[a], x[b]), np.maximum(x[a], x[b])
      x[a], x[b] = lo, hi
    else:
      a, b, c = op
      s = np.sort(np.stack([x[a], x[b], x[c]]), axis=0)
      x[a], x[b], x[c] = s[0], s[1], s[2]
  return x


def check_exhaustive_01(ops, order, n):
  """All 2^n 0/1 inputs, bit-sliced over uint64 words."""
  n_words = max(1, (1 << n) // 64)
  idx = np.arange(n_words, dtype=np.uint64)
  wires = []
  for w in range(n):
    if w < 6:  # pattern inside a word
      pat = 0
      for b in range(64):
        if (b >> w) & 1:
          pat |= 1 << b
      wires.append(np.full(n_words, pat, dtype=np.uint64))
    else:
      wires.append(np.where((idx >> np.uint64(w - 6)) & np.uint64(1),
                            np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64(0)))
  for op in ops:
    if len(op) == 2:
      a, b = op
      wires[a], wires[b] = wires[a] & wires[b], wires[a] | wires[b]
    else:
      a, b, c = op
      x, y, z = wires[a], wires[b], wires[c]
      wires[a], wires[b], wires[c] = (x & y & z, (x & y) | (x & z) | (y & z),
                                      x | y | z)
  for lo, hi in zip(order, order[1:]):   # sorted: never 1 before 0
    if np.any(wires[lo] & ~wires[hi]):
      return False
  return True


def check_merge_sorted_01(merge_fn, lengths):
  """A merge network on every combination of sorted 0/1 lists."""
  net = Net()
  lists, base = [], 0
  for n in lengths:
    lists.append(list(range(base, base + n)))
    base += n
  order = merge_fn(net, *lists)
  cases = list(itertools.product(*[range(n + 1) for n in lengths]))
  x = np.zeros((base, len(cases)), dtype=np.int8)
  for ci, zeros in enumerate(cases):
    for L, z in zip(lists, zeros):
      for k, w in enumerate(L):
        x[w, ci] = 0 if k < z else 1
  y = run(net.ops, x)[order]
  return bool(np.all(np.diff(y.astype(np.int16), axis=0) >= 0))


def build(n_real=50, n_wires=54):
  """The committed scheme: two 27-sorters (3-way merge sort) + one odd-even
  2-way merge, pruned to n_real real inputs."""
  net = Net()
  wires = list(range(n_wires))
  a = sort3k(net, wires[:27])
  b = sort3k(net, wires[27:54])
  order = merge2(net, a, b)
  return prune(net.ops, order, n_wires, n_real)


def emit(path, ops, order, n_real):
  lines = ['// GENERATED by tools/gen_sort3_network.py -- do not edit.',
           f'// Sorting program for exactly {n_real} registers from 2-sorters',
           '// (WB2_S2(a, b): x[a] <= x[b]) and 3-sorters (WB2_S3(a, b, c):',
           '// x[a] <= x[b] <= x[c]; v_min3 / v_med3 / v_max3): two 27-sorters by',
           '// 3-way odd-even merge sort + one 2-way odd-even merge, pruned for the',
           '// +inf padding.  The sorted order is the permutation WB2_SORT3_ORDER_50',
           '// (rank r lives in register order[r]); no data is moved.',
           f'// {sum(len(o) == 2 for o in ops)} 2-sorters + '
           f'{sum(len(o) == 3 for o in ops)} 3-sorters = {cost(ops)} instructions',
           f'// (the pruned Batcher network: 403 comparators = 806).',
           f'#define WB2_SORT3_NETWORK_{n_real} \\']
  body = []
  for i in range(0, len(ops), 6):
    body.append('  ' + ' '.join(
        (f'WB2_S2({o[0]},{o[1]})' if len(o) == 2
         else f'WB2_S3({o[0]},{o[1]},{o[2]})') for o in ops[i:i + 6]))
  lines.append(' \\\n'.join(body))
  lines.append('')
  lines.append(f'#define WB2_SORT3_ORDER_{n_real} \\')
  lines.append('  ' + ', '.join(str(r) for r in order))
  lines.append('')
  open(path, 'w').write('\n'.join(lines))


def main():
  # 1. the 3-way merges on all sorted 0/1 inputs
  for n in (1, 3, 9):
    ok = check_merge_sorted_01(merge3, (n, n, n))
    print(f'merge3 of 3 x {n}: {"ok" if ok else "FAILS"}')
    assert ok
  for la, lb in ((27, 27), (9, 5), (4, 7), (27, 23)):
    ok = check_merge_sorted_01(merge2, (la, lb))
    print(f'merge2 of {la} + {lb}: {"ok" if ok else "FAILS"}')
    assert ok
  # 2. the 27-sorter on all 2^27 0/1 inputs
  net = Net()
  order = sort3k(net, list(range(27)))
  print(f'sort27: {len(net.ops)} ops, {cost(net.ops)} instructions '
        f'(Batcher 32 pruned to 27: '
        f'{cost(prune(*_batcher(32), 32, 27)[0])})')
  ok = check_exhaustive_01(net.ops, order, 27)
  print(f'sort27 on all 2^27 0/1 inputs: {"ok" if ok else "FAILS"}')
  assert ok
  # 3. the pruned 50-input program
  ops, order = build()
  bops, border = prune(*_batcher(64), 64, 50)
  print(f'50 inputs: {cost(ops)} instructions '
        f'({sum(len(o) == 3 for o in ops)} 3-sorters, '
        f'{sum(len(o) == 2 for o in ops)} 2-sorters); Batcher 64 pruned: '
        f'{cost(bops)}')
  rs = np.random.RandomState(0)
  x = rs.standard_normal((50, 20000)).astype(np.float32)
  x[:, :2000] = np.round(x[:, :2000] * 2) / 2        # many ties
  y = run(ops, x)[order]
  assert np.array_equal(y, np.sort(x, axis=0)), 'pruned program does not sort'
  print('pruned program sorts 20000 random columns (with ties): ok')
  if '--emit' in sys.argv:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, 'weatherbench2_amd', 'csrc',
                        'sort3_network_50.inc')
    emit(path, ops, order, 50)
    print('wrote', path)


def _batcher(n):
  net = Net()
  order = batcher_sort(net, list(range(n)))
  return net.ops, order


if __name__ == '__main__':
  main()
