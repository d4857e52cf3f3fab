"""The bench line's driver-facing shape (CPU: no GPU needed).  The driver keeps the scalar fields of `roofline` and the head of the
line; the north-star kernel's figures, the encoders' fractions and the main-pass kernel's register numbers must therefore be scalars at
the top level and inside `roofline` (r04 verdict item 7).  Also: the main-pass kernel of the SHIPPED library spills nothing — read from
the metadata note of the device code inside libfsgpu.so, the way bench.py reports it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_scalars_reach_the_top_level_and_the_roofline_object():
    import bench
    line = {
        "metric": "m", "value": 1.0, "unit": "queries/sec", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 3.0,
        "higher_is_better": True, "p50_phase1_latency_ms": 1.5, "p50_phase0_latency_ms": 0.5, "p50_latency_ms_single_query": 0.7,
        "end_to_end_queries_per_sec": 9.0,
        "roofline": {"bound": "mfma", "frac": 0.5, "hbm": {"frac": 0.3}, "joint": {"frac": 0.4},
                     "exact_f16_scan": {"frac": 0.77, "avg_launch_ms": 1.24, "achieved": 6100.0, "traffic": 7.7e9, "algorithmic_bytes_per_launch": 7.68e9}},
        "encoders": {"minilm_l6": {"gpu_ms_per_batch": 0.33, "roofline": {"frac": 0.12},
                                   "documents_32x512": {"gpu_ms_per_batch": 0.87, "roofline": {"frac": 0.2}}}},
    }
    bench.flatten_for_the_driver(line)
    keys = list(line.keys())
    assert keys[:7] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step"]
    for k in ("p50_latency_ms_single_query", "p50_phase1_latency_ms", "p50_phase0_latency_ms", "end_to_end_queries_per_sec"):
        assert keys.index(k) < keys.index("higher_is_better"), k
    assert line["roofline_exact_f16_frac"] == 0.77 and line["roofline"]["exact_f16_frac"] == 0.77
    assert line["roofline_exact_f16_ms"] == 1.24 and line["roofline_exact_f16_traffic"] == 7.7e9
    assert line["roofline_encoder_queries_mfma_frac"] == 0.12 and line["roofline_encoder_documents_ms"] == 0.87
    assert line["roofline_joint_frac"] == 0.4 and line["roofline_main_pass_hbm_frac"] == 0.3
    assert all(not isinstance(v, (dict, list)) for k, v in line.items() if k.startswith("roofline_"))
    assert keys.index("roofline_exact_f16_frac") < keys.index("roofline")


def test_the_shipped_main_pass_kernel_spills_nothing():
    import bench
    from __graft_entry__ import build
    build()
    res = bench.main_pass_kernel_resources()
    assert res is not None, "scan_wide_kernel<384, 1, 4, 3, 30, 0> not found in libfsgpu.so"
    assert res["vgpr_spill_count"] == 0 and res["private_segment_fixed_size"] == 0, res
    assert res["vgpr_count"] <= 256, res


def test_default_layout_of_an_n_gpu_run():
    import bench
    assert [bench.default_query_groups(n) for n in (1, 2, 3, 4, 6, 8)] == [1, 2, 1, 2, 2, 2]
