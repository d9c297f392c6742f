"""bench.py's algorithmic flops/sample constants are the oracle's operation counters (SURVEY §8 d
conventions); re-derive the headline one here so the roofline numerator cannot drift silently."""
import importlib.util
import os

from oracle_lib import OracleScene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_flops_per_sample_matches_oracle_counters():
    b = _bench()
    for name, w, spp in (("cornell_box", 96, 16), ("cornell_smoke", 64, 8)):
        _, cnt = OracleScene(name, 1.0).render(w, w, 0, spp, threads=8)
        fps = cnt["flops"] / cnt["paths"]
        assert abs(fps - b.FLOPS_PER_SAMPLE[name]) / b.FLOPS_PER_SAMPLE[name] < 0.05, (name, fps)


def test_peaks_are_the_guides():
    b = _bench()
    assert b.PEAK_TFLOPS["f32"] == 157.3 and b.PEAK_TFLOPS["f64"] == 78.6 and b.PEAK_HBM_GBS == 8000.0
    # 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
    assert abs(256 * 4 * 32 * 2 * 2.4e9 / 1e12 - b.PEAK_TFLOPS["f32"]) < 0.1
