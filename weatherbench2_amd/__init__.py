"""weatherbench2_amd: MI355X-native per-chunk metric evaluation for WeatherBench 2.

Drop-in operators for the reference's `Metric.compute_chunk` / `config.Eval`
API backed by hand-written HIP kernels (libwb2hip.so, C ABI in include/wb2hip.h).
"""
__version__ = '0.1.0'
