"""absltest subset on top of unittest."""
import unittest


class TestCase(unittest.TestCase):

  def assertLen(self, container, expected_len, msg=None):
    self.assertEqual(len(container), expected_len, msg)

  def assertEmpty(self, container, msg=None):
    self.assertEqual(len(container), 0, msg)

  def assertBetween(self, value, lo, hi, msg=None):
    self.assertTrue(lo <= value <= hi, msg or f'{value} not in [{lo}, {hi}]')

  def assertSameElements(self, a, b, msg=None):
    self.assertEqual(set(a), set(b), msg)


def main(*args, **kwargs):
  unittest.main(*args, **kwargs)
