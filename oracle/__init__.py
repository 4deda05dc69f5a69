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

Parity that NO reference test pins to a number (MSE/MAE/Bias/ACC values on
random data, slice regions other than the tropics, float32 inputs, spectrum
absolute values) is "parity unpinned": for those the oracle *is* the
definition and is kept line-by-line traceable to the cited reference lines.
"""
