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


# ---------------------------------------------------------------------------
# Round 4: programs for the other ensemble sizes K3 instantiates exactly
# (csrc/sort3_networks.inc): the cheapest of a few constructions per size.
# ---------------------------------------------------------------------------
# member counts K3 instantiates with a compile-time M (50 has its own file):
# small test ensembles, 10 / 11 / 20 / 21 / 31 (operational centres with and
# without the control), 25, 30 (the 1990-2019 probabilistic climatology), 32,
# 40, 51 / 56 (IFS ENS + control, GenCast-style), 64, 100 -- and 36, 45, 48,
# 72, 80, 90 so that every other count up to 100 finds a HOST program at most
# ~10 % larger than itself (ens_point_hosted: dead slots at +inf)
EXACT_SIZES = (4, 5, 8, 10, 11, 16, 20, 21, 25, 30, 31, 32, 36, 40, 45, 48,
               51, 56, 64, 72, 80, 90, 100)


def _sorted_by(builder, n_wires, n_real):
  net = Net()
  order = builder(net, list(range(n_wires)))
  return prune(net.ops, order, n_wires, n_real)


def _blocks_of_27(net, wires):
  """27-sorters on consecutive blocks, merged pairwise (2-way odd-even)."""
  parts = [sort3k(net, wires[i:i + 27]) for i in range(0, len(wires), 27)]
  while len(parts) > 1:
    nxt = [merge2(net, parts[i], parts[i + 1]) if i + 1 < len(parts)
           else parts[i] for i in range(0, len(parts), 2)]
    parts = nxt
  return parts[0]


_BEST: dict = {}


def best_program(n):
  """(ops, order) sorting n registers: the cheapest of
    * the 3-way merge sort on 3^k blocks + 2-way merges (`sort_general`),
    * k 27-sorters merged pairwise, pruned to n,
    * Batcher's network on the next power of two, pruned to n,
    * the best programs of two parts merged (every split, recursively)."""
  if n in _BEST:
    return _BEST[n]
  cands = []
  if n <= 2:
    cands.append(_sorted_by(sort_general, n, n))
  else:
    cands.append(_sorted_by(sort_general, n, n))
    p = 1
    while p < n:
      p *= 2
    cands.append(prune(*_batcher(p), p, n))
    for blocks in (1, 2, 3):
      if 27 * (blocks - 1) < n <= 27 * blocks:
        cands.append(_sorted_by(_blocks_of_27, 27 * blocks, n))
    p3 = 3
    while p3 < n:
      p3 *= 3
    if p3 <= 81:
      cands.append(_sorted_by(sort3k, p3, n))
    for a in range(max(1, n // 2 - 6), n // 2 + 1):  # near-even splits
      oa, ra = best_program(a)
      ob, rb = best_program(n - a)
      net = Net()
      net.ops = list(oa) + [tuple(w + a for w in op) for op in ob]
      order = merge2(net, list(ra), [w + a for w in rb])
      # two sorted parts + a merge that is correct on every pair of sorted 0/1
      # lists of these lengths (0-1 principle) = a sorter
      assert check_merge_sorted_01(merge2, (a, n - a))
      cands.append((net.ops, order))
  _BEST[n] = min(cands, key=lambda c: cost(c[0]))
  return _BEST[n]


def check_program(ops, order, n, rs):
  x = rs.standard_normal((n, 20000)).astype(np.float32)
  x[:, :4000] = np.round(x[:, :4000] * 2) / 2        # many ties
  x[:, 4000:4100] = np.float32(np.inf) * (rs.rand(n, 100) < 0.2)
  x[np.isnan(x)] = 0
  assert sorted(order) == list(range(n)), 'order is not a permutation'
  assert np.array_equal(run(ops, x)[order], np.sort(x, axis=0))
  if n <= 24:
    assert check_exhaustive_01(ops, order, n), f'0-1 principle fails, n={n}'
    return 'all 2^%d 0/1 inputs' % n
  # larger: 2^22 random 0/1 inputs, bit-sliced, + every sorted-halves input
  words = rs.randint(0, 2**63, size=(n, 1 << 16), dtype=np.int64).astype(
      np.uint64)
  w = [words[i].copy() for i in range(n)]
  for op in ops:
    if len(op) == 2:
      a, b = op
      w[a], w[b] = w[a] & w[b], w[a] | w[b]
    else:
      a, b, c = op
      x0, y0, z0 = w[a], w[b], w[c]
      w[a], w[b], w[c] = (x0 & y0 & z0, (x0 & y0) | (x0 & z0) | (y0 & z0),
                          x0 | y0 | z0)
  for lo, hi in zip(order, order[1:]):
    assert not np.any(w[lo] & ~w[hi]), f'0-1 sample fails, n={n}'
  return '2^22 sampled 0/1 inputs (blocks and merges verified exhaustively)'


def emit_exact(path):
  rs = np.random.RandomState(1)
  lines = ['// GENERATED by tools/gen_sort3_network.py --emit-exact -- do not edit.',
           '// Sorting programs from 2-sorters (WB2_S2(a, b): x[a] <= x[b]) and',
           '// 3-sorters (WB2_S3(a, b, c): v_min3 / v_med3 / v_max3) for the ensemble',
           '// sizes K3 instantiates with a compile-time member count; rank r ends up',
           '// in register WB2_SORT3_ORDER_<M>[r], no data is moved.', '']
  def npad(n):
    p = 1
    while p < n:
      p *= 2
    return max(p, 4)
  lines.append('// X(member count, padded register count) of every program below')
  lines.append('#define WB2_SORT3_SIZES(X) \\')
  lines.append('  ' + ' '.join(f'X({n}, {npad(n)})' for n in EXACT_SIZES))
  lines.append('')
  for n in EXACT_SIZES:
    ops, order = best_program(n)
    how = check_program(ops, order, n, rs)
    p = 1
    while p < n:
      p *= 2
    bops, _ = prune(*_batcher(p), p, n)
    print(f'{n} members: {cost(ops)} instructions '
          f'({sum(len(o) == 3 for o in ops)} 3-sorters, '
          f'{sum(len(o) == 2 for o in ops)} 2-sorters); pruned Batcher '
          f'{cost(bops)}; verified on {how}')
    lines.append(f'// {n}: {sum(len(o) == 2 for o in ops)} 2-sorters + '
                 f'{sum(len(o) == 3 for o in ops)} 3-sorters = {cost(ops)} '
                 f'instructions (pruned Batcher: {cost(bops)})')
    lines.append(f'#define WB2_SORT3_NETWORK_{n} \\')
    body = []
    for i in range(0, len(ops), 6):
      body.append('  ' + ' '.join(
          (f'WB2_S2({o[0]},{o[1]})' if len(o) == 2
           else f'WB2_S3({o[0]},{o[1]},{o[2]})') for o in ops[i:i + 6]))
    lines.append(' \\\n'.join(body))
    lines.append(f'#define WB2_SORT3_ORDER_{n} \\')
    lines.append('  ' + ', '.join(str(r) for r in order))
    lines.append('')
  open(path, 'w').write('\n'.join(lines))
  print('wrote', path)


if __name__ == '__main__' and '--emit-exact' in sys.argv:
  emit_exact(os.path.join(
      os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
      'weatherbench2_amd', 'csrc', 'sort3_networks.inc'))
