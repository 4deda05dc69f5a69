"""Import-only stand-in for absl (absent here) so that the reference's own
*_test.py files run under unittest.  TEST INFRASTRUCTURE ONLY."""
