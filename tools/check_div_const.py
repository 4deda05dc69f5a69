"""Exhaustive check: x / C in float32 as one multiply and two FMAs.

  python tools/check_div_const.py 49 50

For a compile-time divisor C the ensemble kernel (csrc/ensemble.hip div_const)
replaces the 10-instruction IEEE division by Markstein's sequence

    r  = RN(1 / C)             (compile time)
    q0 = RN(x * r)
    e  = RN(x - C * q0)        (one FMA; exact when it does not underflow)
    q1 = RN(q0 + e * r)        (one FMA)

q1 must equal RN(x / C) for EVERY float32 x.  The sequence commutes with
scaling by powers of two as long as nothing leaves the normal range (the kernel
takes the IEEE division whenever e is not a normal number or zero), so it is
enough to check all 2^23 mantissas of one binade -- in exact integer
arithmetic, no floating point involved:  x = m, 2^23 <= m < 2^24.
"""
import sys

import numpy as np


def rne_shift(n, shift):
  """round-to-nearest-even of n / 2^shift (n >= 0, int64 arrays)."""
  q = n >> shift
  rem = n - (q << shift)
  half = np.int64(1) << (shift - 1)
  up = (rem > half) | ((rem == half) & ((q & 1) == 1))
  return q + up


def to24(n):
  """n > 0 (int64) -> (Q, s): Q = RNE(n / 2^s) with 2^23 <= Q <= 2^24."""
  bits = np.floor(np.log2(n.astype(np.float64))).astype(np.int64) + 1
  # log2 of a float64-rounded n can be off by one at powers of two: fix up
  bits = np.where((np.int64(1) << (bits - 1)) > n, bits - 1, bits)
  bits = np.where((np.int64(1) << bits) <= n, bits + 1, bits)
  s = bits - 24
  assert (s > 0).all()
  return rne_shift(n, s), s


def check(c):
  # r = RN24(1 / c) = R * 2^-k with 2^23 <= R < 2^24
  k = 23
  while (1 << k) // c < (1 << 23):
    k += 1
  num = 1 << k
  R = num // c
  rem = num - R * c
  if 2 * rem > c or (2 * rem == c and R & 1):
    R += 1
  assert (1 << 23) <= R < (1 << 24)
  m = np.arange(1 << 23, 1 << 24, dtype=np.int64)
  # q0 = RN(m * R * 2^-k) = Q0 * 2^(s0 - k)
  Q0, s0 = to24(m * R)
  # e = m - c * q0 = (m * 2^(k - s0) - c * Q0) * 2^(s0 - k): exact, small
  E = (m << (k - s0)) - c * Q0
  assert np.abs(E).max() < (1 << 24)  # representable
  # q1 = RN(q0 + e * r) = RN((Q0 * 2^k + E * R) * 2^(s0 - 2k))
  N = (Q0 << k) + E * R
  Q1, s1 = to24(N)
  # got = Q1 * 2^(s1 + s0 - 2k);  want = RN24(m / c)
  # want: scale m so that the quotient has 24+ bits: m * 2^j / c
  j = 30
  num = m << j
  qf = num // c
  rf = num - qf * c  # exact remainder -> sticky
  # qf has ~47..48 bits; round to 24 with the remainder as a sticky bit
  bits = np.floor(np.log2(qf.astype(np.float64))).astype(np.int64) + 1
  bits = np.where((np.int64(1) << (bits - 1)) > qf, bits - 1, bits)
  bits = np.where((np.int64(1) << bits) <= qf, bits + 1, bits)
  s = bits - 24
  q = qf >> s
  low = qf - (q << s)
  half = np.int64(1) << (s - 1)
  up = (low > half) | ((low == half) & (rf > 0)) | \
       ((low == half) & (rf == 0) & ((q & 1) == 1))
  W = q + up
  # compare values: Q1 * 2^(s1 + s0 - 2k)  vs  W * 2^(s - j)
  ea = s1 + s0 - 2 * k
  eb = s - j
  # normalise a possible carry to 2^24
  carry = Q1 == (1 << 24)
  Q1 = np.where(carry, Q1 >> 1, Q1)
  ea = np.where(carry, ea + 1, ea)
  carry = W == (1 << 24)
  W = np.where(carry, W >> 1, W)
  eb = np.where(carry, eb + 1, eb)
  bad = (Q1 != W) | (ea != eb)
  return int(bad.sum()), m[bad][:5]


if __name__ == '__main__':
  for c in [int(a) for a in sys.argv[1:]] or [49, 50]:
    n_bad, examples = check(c)
    print(f'C = {c}: {n_bad} of {1 << 23} mantissas misrounded', examples)
