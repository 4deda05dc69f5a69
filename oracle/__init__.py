"""CPU oracle for the WeatherBench2 metric-evaluation hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain NumPy/SciPy restatement of
the reference algorithm (``/root/reference/weatherbench2/metrics.py``,
``regions.py``, ``derived_variables.py:531-626``, ``evaluation.py:388-438``)
and is the *checker* for the HIP path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under ``weatherbench2_amd/`` does.

Why a restatement and not the reference itself: the reference is pure Python
on top of ``xarray`` / ``apache_beam`` / ``xarray_beam``, none of which are
installed in this image (no network), so ``import weatherbench2`` fails.  The
arithmetic lives in un-vendored third-party code (xarray >= 2024.11
``DatasetWeighted.mean`` -> ``np.einsum``; numpy >= 2.1.3 ``argsort`` /
``fft.rfft``).  The oracle restates that published
behaviour and is pinned by the reference's own known-answer tests, ported in
``tests/test_oracle_golden.py`` (lat weights, wind-vector RMSE ``[0, 10,
nan]``, NaN/Inf region masking, ``_rankdata`` == scipy ordinal ranks, CRPS ==
brute-force eFAIR, land-region masking, Parseval and spectral-peak tests).

Second pin (round 2): the reference's OWN code does run here once `xarray`
resolves to the NumPy stand-in of ``oracle/refshim`` (a restatement of the
xarray semantics the reference touches, itself checked by running the
reference's own 82 unit tests on it: ``oracle/refshim/run_reference_tests.py``).
``tests/golden/make_reference_vectors.py`` imports the reference's unmodified
``metrics.py`` / ``regions.py`` / ``thresholds.py`` / ``derived_variables.py``
/ ``evaluation.py`` and records their outputs on seeded inputs
(``tests/golden/reference_vectors_v1.npz``); the oracle reproduces every one of
them to 1e-12 (``tests/test_reference_vectors.py``), which covers what no
reference test pins to a number: MSE/MAE/Bias/ACC on random data, every region
type, float32 inputs and float32 coordinates, the by-init layout, spectrum
absolute values, the tier-2 metrics and the seeded rank histogram.  What stays
restated rather than run is xarray itself (the stand-in), NumPy/SciPy being the
real libraries.
"""
