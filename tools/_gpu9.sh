cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - <<'EOF'
import sys
sys.path.insert(0,'.')
import numpy as np
from tests import helpers, official_chunks as oc
from tests.test_chunk_program_gpu import _setup
from weatherbench2_amd import evaluation, program
_, _, gf, gt, cfg = _setup(n_init=3, n_lead=2, n_lat=31, n_lon=72)
chunks = oc.chunk_pairs(gf, gt)
try:
  out = evaluation.evaluate_chunks(chunks, cfg, False, prefetch=0, batch_chunks=1)
except Exception as e:
  import traceback; traceback.print_exc()
print('REASONS', program.REASONS)
EOF
