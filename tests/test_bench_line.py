"""The ONE stdout line of bench.py must fit the driver's stdout tail (round 4: a 21.8 KB line was cut at the front and the
driver recorded `parsed: null`).  CPU only: synthetic KernelTimer summaries through the same assembly functions."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _summary(n_groups=40, n_shapes=24):
    summ = {}
    for g in range(n_groups):
        conv = g % 4 == 0
        name = (f"conv3x3_kind{g}_unsupervised" if conv else f"some_hot_path_kernel_{g}_unsupervised")
        shapes = {f"B8x{128 + s}->128x109x256d{s % 5}": {"launches": 4, "ms_total": 0.4 + 0.01 * s, "bytes": 1.0e8, "flops": 2.0e11}
                  for s in range(n_shapes)}
        summ[name] = {"launches": 4 * n_shapes, "ms_total": sum(v["ms_total"] for v in shapes.values()), "bytes_total": 1.0e8 * 4 * n_shapes,
                      "flops_total": 2.0e11 * 4 * n_shapes, "by_shape": shapes}
    return summ


def _line(kernels):
    per_level = {f"B8xC{c}x{h}x{w}": {"launches": 10, "avg_us": 11.64, "GBps": 2042.5}
                 for c, h, w in ((196, 7, 16), (128, 14, 32), (96, 28, 64), (64, 55, 128), (32, 109, 256))}
    roofline = {"bound": "hbm", "achieved": 1957.1, "peak": 8000.0, "unit": "GB/s", "frac": 0.2446, "traffic": 20142951,
                "kernel": "ocf::corr_fwd_rows (levels 2-3) / corr_fwd_wave (levels 4-6) (ocf_corr_fwd)", "launches": 50, "avg_launch_us": 10.28,
                "algorithmic_bytes_per_launch": 20112640, "tflops": 40.22, "per_level": per_level, "event_regime": "x" * 120,
                "traffic_source": "y" * 115, "rocprof_avg_launch_us": 9.19, "kernels": kernels,
                "fused_op": {"bound": "hbm", "achieved": 1350.9, "peak": 8000.0, "unit": "GB/s", "frac": 0.1689, "operator": "z" * 100,
                             "instances": {"one_launch": 32, "warp_then_correlation": 0, "no_flow (top level)": 8}, "source": "s" * 130},
                "copy_ceiling_GBps": 4778.3, "kernels_detail": "profiles/bench_kernels_latest.json"}
    leg = {"metric": "m" * 60, "value": 869.144, "unit": "image-pairs/s", "ms_per_step": 9.204, "steps": 10, "launch": "hipGraph replay",
           "final_loss": 0.22775, "workload": "w" * 90}
    cpu = {"value": 2.729, "unit": "image-pairs/s", "cores": 16, "kind": "port", "cpu_model": "AMD EPYC 9575F 64-Core Processor", "best": 2.8,
           "median": 2.729, "sample": "s" * 200, "compute_cost_volume_levels_b1_fp32": {f"C{c}": {"best_ms": 1.234, "GBps": 0.612} for c in range(5)},
           "cfg3_pwc_b8_384x512": {"pairs_per_s": 2.31, "best_pairs_per_s": 2.4, "s_per_step": 3.46, "steps": 3, "note": "n" * 40}}
    return {"metric": "image-pairs/sec at 436x1024 bf16 (training step)", "value": 871.0, "unit": "image-pairs/s", "n_gpus": 1, "steps": 20,
            "warmup": 5, "ms_per_step": 9.19, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic", "config": {"workload": "c" * 200, "global_batch": 8, "height": 436, "width": 1024, "layout": "cl",
                                            "parallelism": "dp1", "launch": "hipGraph replay of the whole step", "final_loss": 3.17,
                                            "optimizer": "adam(fused)", "weights": "fp32 masters + bf16 compute copies", "network_execution": "e" * 120},
            "roofline": roofline, "unsupervised": leg, "cfg3": dict(leg), "cpu_baseline": cpu}


def test_forty_kernel_groups_fit_the_driver_tail():
    detail = bench.kernel_rooflines(_summary(40, 24), bench.HBM_PEAK_GBPS)
    assert len(detail) == 40 and len(json.dumps(detail)) > 20000  # the per-shape tables are what broke round 4
    line = bench.fit_line(_line(bench.compact_kernels(detail)))
    text = json.dumps(line)
    assert len(text) < 8000
    back = json.loads(text)
    assert "dropped_for_length" not in back  # 40 groups fit without dropping anything
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in back
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(back["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(back["cpu_baseline"])
    assert len(back["roofline"]["kernels"]) == 40 and all(isinstance(v, float) for v in back["roofline"]["kernels"].values())


def test_an_oversized_line_sheds_detail_but_never_the_contract():
    detail = bench.kernel_rooflines(_summary(400, 2), bench.HBM_PEAK_GBPS)
    line = bench.fit_line(_line(bench.compact_kernels(detail)))
    text = json.dumps(line)
    assert len(text) < 8000
    back = json.loads(text)
    assert "roofline.kernels" in back["dropped_for_length"]
    assert back["roofline"]["frac"] == 0.2446 and back["cpu_baseline"]["value"] == 2.729 and back["value"] == 871.0
