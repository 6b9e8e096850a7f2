"""CPU: the algorithmic-work figures that bench.py's `roofline` object and DESIGN.md price kernels against
(BASELINE.md / SURVEY.md section 8d) follow from the architecture configuration: scripts/roofline.py recomputes them."""
import importlib.util
from pathlib import Path

from seamless_communication_amd.config import seamless_m4t_v2_large

spec = importlib.util.spec_from_file_location("roofline", Path(__file__).resolve().parent.parent / "scripts" / "roofline.py")
roofline = importlib.util.module_from_spec(spec)
spec.loader.exec_module(roofline)


def test_quoted_figures_follow_from_the_config():
    w = roofline.work(seamless_m4t_v2_large(), frames=998, text_len=41, units=500)
    e, d, t, v = w["encoder+adaptor"], w["text_decoder_per_step"], w["t2u"], w["vocoder"]
    assert (e["S"], e["S_a"]) == (499, 63)
    assert abs(e["params"] / 1e6 - 635.0) < 0.05 and abs(e["weight_bytes"] / 1e9 - 1.270) < 0.001
    assert abs(e["flops_reference_formulation"] / 1e9 - 629.3) / 629.3 < 0.005  # the survey's count (with the rel-pos einsum)
    assert e["flops"] < e["flops_reference_formulation"]  # the q.R table needs less than the reference's einsum
    assert abs((d["layer_params"] + d["projection_params"]) / 1e6 - 866.7) < 0.2
    assert abs(d["layer_params"] / 1e6 - 604.5) < 0.2 and abs(d["weight_bytes"] / 1e9 - 1.733) < 0.002
    assert abs(w["text_decoder_cross_kv_precompute"]["flops"] / 1e9 - 6.34) < 0.01
    assert abs(t["encoder_params"] / 1e6 - 125.9) < 0.1 and abs(t["decoder_params"] / 1e6 - 113.3) < 0.1
    assert abs(t["projection_params"] / 1e6 - 10.3) < 0.05 and abs(t["flops"] / 1e9 - 140) < 2
    assert abs(v["flops"] / 1e9 - 166.3) < 0.5 and v["output_samples"] == 160000
