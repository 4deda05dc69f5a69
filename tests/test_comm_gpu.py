"""wb2_time_mean_allreduce / wb2_comm_*: the path's one exchange step through
the C ABI (RCCL opened by libwb2hip.so itself, no torch.distributed).  A gpurun
box has one GPU, so the communicator has ONE rank: the all-reduce must then be
the identity, which still exercises library loading, communicator set-up, the
grouped in-place call on the caller's stream and teardown.  The N > 1 arithmetic
is covered by the gloo tests (tests/test_distributed_cpu.py) and bench.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_allreduce_through_the_c_abi():
  import torch
  from weatherbench2_amd import engine, evaluation
  from weatherbench2_amd import xarray_lite as xl
  dev = torch.device('cuda')
  comm = engine.comm_init_rank(engine.comm_unique_id(), 1, 0)
  try:
    gen = torch.Generator(device=dev).manual_seed(0)
    total = torch.randn(5 * 13 * 13, dtype=torch.float64, device=dev,
                        generator=gen)
    count = torch.full_like(total, 7.0)
    want_t, want_c = total.clone(), count.clone()
    engine.time_mean_allreduce(total, count, comm)
    torch.cuda.synchronize()
    assert torch.equal(total, want_t) and torch.equal(count, want_c)
    # RunningMean with the RCCL communicator == RunningMean without
    rs = np.random.RandomState(1)
    chunks = []
    for _ in range(3):
      a = rs.standard_normal((2, 3, 2, 4))
      a[rs.rand(*a.shape) < 0.1] = np.nan
      chunks.append(xl.Dataset(
          {'z': xl.DataArray(a, ('metric', 'region', 'init_time', 'lead'))},
          {'metric': np.array(['a', 'b'], dtype=object),
           'region': np.array(['r0', 'r1', 'r2'], dtype=object)}))
    with_comm = evaluation.RunningMean('init_time', True, dev, comm=comm)
    plain = evaluation.RunningMean('init_time', True, dev)
    for c in chunks:
      with_comm.add(c)
      plain.add(c)
    np.testing.assert_array_equal(with_comm.result()['z'].values,
                                  plain.result()['z'].values)
  finally:
    engine.comm_destroy(comm)
