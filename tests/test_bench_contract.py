"""bench.py contract on the CPU: the single-GPU leg runs end to end against the host test double at a tiny scale
and emits the fields the driver and the judge read (metric/value/unit/..., roofline, cpu_baseline, parity)."""
import argparse
import json

import bench


def test_single_gpu_leg_fields(host_engine):
    args = argparse.Namespace(gpus=1, steps=2, warmup=1, scale=10, ef=8, no_cpu=False, no_secondary=False, no_symmetric=False)
    out = bench.single_gpu(args)
    line = json.loads(json.dumps(out))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["unit"] == "GTEPS" and line["n_gpus"] == 1 and line["dtype"] == "f32" and line["data"] == "synthetic"
    assert line["vs_baseline"] is None and line["higher_is_better"] is True
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["peak"] == 8000.0
    assert line["parity"]["rel_linf"] <= 1e-6
    assert line["parity"]["gpu_iterations"] == line["parity"]["cpu_iterations"]
    assert set(line["secondary"]) == {"ppr_mabs_default_tol1e-6", "ppr_50_iterations", "heat_kernel_t5_31_iterations",
                                      "heat_kernel_t5_31_iterations_chebyshev", "absorbing_walks_a085_l1_1e-6",
                                      "ppr_l1_1e-6_symmetrised_graph", "ppr_l1_1e-6_batch_of_64_seeds", "ppr_l1_1e-6_real_weights",
                                          "ppr_l1_1e-6_backend_primitives", "heat_kernel_t5_31_iterations_backend_primitives",
                                          "absorbing_walks_a085_l1_1e-6_backend_primitives", "ppr_l1_1e-9_f64_iterates"}
    prim = line["secondary"]["ppr_l1_1e-6_backend_primitives"]
    assert "error" not in prim and prim["spmv_per_run"] > 0, prim
    assert set(("graph_build_s", "graph_rebuild_s", "graph_build_ms")) <= set(line["config"])
    weighted = line["secondary"]["ppr_l1_1e-6_real_weights"]
    assert "error" not in weighted and weighted["gteps"] >= 0 and weighted["iterations"] > 2, weighted
    batch = line["secondary"]["ppr_l1_1e-6_batch_of_64_seeds"]
    assert "error" not in batch and batch["width"] == 64 and batch["edge_vector_products_per_s_G"] > 0
    assert line["secondary"]["ppr_50_iterations"]["spmv_per_run"] == 50
    assert line["roofline"]["frac_of_measured_copy"] is not None and "residual" in line["roofline"]["kernel"]
    assert line["cpu_baseline"]["all_cores"]["cores"] >= 1
    assert line["secondary"]["heat_kernel_t5_31_iterations"]["spmv_per_run"] in (29, 30)
