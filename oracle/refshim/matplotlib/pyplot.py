def __getattr__(name):
  raise NotImplementedError('wb2shim: matplotlib is not available')
