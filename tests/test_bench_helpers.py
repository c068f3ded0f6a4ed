"""bench.py's bookkeeping without a GPU: the roofline objects must be recomputable from the files under profiles/ (the newest
round's kernel model, PMC summary and rocprofv3 kernel statistics), and the CPU baseline must be sized by the cores the process
may actually use."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_profile_files_bench_reads():
    model, src = bench.load_profile_json("kernel_model.json")
    assert model and src.startswith("profiles/r") and src.endswith("_kernel_model.json")
    # the static instruction count of the committed kernel and the measured issue rates, with their sources
    assert 2000 < model["mads_per_addition"] < model["instr_per_addition"] < 10000
    assert 1e13 < model["mad_peak_per_s"] < 1e14 and 1e9 < model["addition_register_loop_per_s"] < 1e11
    assert "count_accum_instrs" in model["instr_source"] and "microbench" in model["rate_source"]
    # the ceiling of the MAD fraction for this instruction mix (a MAD costs ~2.3x a plain VALU instruction): what int_roofline.frac
    # can reach if the kernel ran at the register-only rate
    ceiling = model["addition_register_loop_per_s"] * model["mads_per_addition"] / model["mad_peak_per_s"]
    assert 0.5 < ceiling < 0.7
    pmc, psrc = bench.load_profile_json("pmc_msm.json")
    assert pmc["msm_n"] == 1 << 20 and pmc["k_bucket_accum"]["hbm_bytes_per_launch"] > 128 * (1 << 20)
    assert psrc.split("_")[0] == src.split("_")[0]          # the same round's collection
    ms, rsrc = bench.rocprof_avg_ms("k_bucket_accum")
    assert 1.0 < ms < 10.0 and rsrc.endswith("_msm_only_kernel_stats.csv")
    assert bench.rocprof_avg_ms("no_such_kernel")[0] is None


def test_committed_bench_line_is_self_consistent():
    """the round's committed line: every roofline fraction recomputes from the numbers beside it"""
    j, _ = bench.load_profile_json("bench.json")
    r = j["roofline"]
    assert abs(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9 - r["achieved"]) < 0.05
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-4 and r["frac"] < 0.05 and r["traffic"] > r["algorithmic_bytes_per_launch"]
    i = j["int_roofline"]
    adds_per_s = i["additions_per_launch"] / (r["avg_launch_ms"] * 1e-3)
    assert abs(adds_per_s * i["mads_per_addition"] / 1e12 - i["achieved"]) < 0.02 and abs(i["achieved"] / i["peak"] - i["frac"]) < 1e-3
    n = j["roofline_ntt"]
    assert n["algorithmic_bytes"] == 288 * n["M"] and abs(n["algorithmic_bytes"] / (n["ms_per_product"] * 1e-3) / 1e9 / n["peak"] - n["frac"]) < 1e-3
    p = j["roofline_prove"]
    nn, Q = j["config"]["n"], j["config"]["Q"]
    assert p["scalar_muls_per_proof"] == 27 * nn + 28 + 2 * Q + Q * (11 * nn + Q)
    c = j["cpu_baseline"]
    assert c["at_bench_size"] and c["n"] == nn and c["same_bytes_as_gpu_proof"] is True and c["cores"] >= 1
    assert j["value"] > 0 and abs(j["n_gpus"] * 1e3 / j["ms_per_step"] - j["value"]) / j["value"] < 0.01
    assert j["msm_strong"]["scaling"] == "strong" and j["scaling"] == "weak" and j["vs_baseline"] is None


def test_round6_legs_in_the_committed_line():
    """round 6: BASELINE configs[1] and configs[4], the reference's criterion shape, the dense-weights
    sensitivity and the integer roof of a whole proof are legs of the SAME line, and their numbers hang together"""
    j, src = bench.load_profile_json("bench.json")
    if int(src.split("/")[1][1:3]) < 6:
        import pytest
        pytest.skip("no round-6 collection published yet")
    c2, c5, cr = j["config2"], j["config5"], j["criterion_shape"]
    assert c2["n"] == 1 << 14 and c2["d"] == 8 * c2["n"] and c2["same_bytes_streamed_and_sequential"] is True
    assert abs(1e3 / c2["streamed"]["ms_per_proof"] - c2["streamed"]["proofs_per_s"]) / c2["streamed"]["proofs_per_s"] < 0.01
    assert c2["scalar_muls_executed_per_proof"] == bench.scalar_muls_executed(c2["n"], c2["Q"], True)
    assert c5["n"] == 1 << 16 and c5["proofs"] == 64 and c5["same_bytes_as_one_handle_alone"] is True
    assert c5["assignment_resident"]["proofs_per_s_per_gpu"] >= c5["assignment_per_proof_from_host"]["proofs_per_s_per_gpu"] * 0.9
    for k in ("example_n1_Q2", "example_n2_Q5"):
        assert cr[k]["verified"] is True and cr[k]["d"] == 25 * cr[k]["n"] and 0 < cr[k]["prove_ms"] < cr[k]["srs_new_plus_prove_ms"]
    dw = j["sensitivities"]["dense_weights"]
    assert dw["resident_unprepared"]["same_bytes_as_prepared"] is True and dw["one_shot"]["same_bytes_as_streamed"] is True
    ip, model = j["int_roofline_prove"], bench.load_profile_json("kernel_model.json")[0]
    adds = ip["walk_additions_per_proof"] + ip["reduction_additions_per_proof_walk_equivalents"]
    assert abs(adds * j["value"] / j["n_gpus"] * model["mads_per_addition"] / 1e12 - ip["achieved"]) / ip["achieved"] < 0.02
    assert abs(ip["achieved"] / ip["peak"] - ip["frac"]) < 1e-3 and ip["frac"] < j["int_roofline"]["frac"] + 0.05
    e = j["msm_strong"]["emulated_share"]
    assert e["best_mode"] in ("bucket_ranges", "term_ranges") and e["best_speedup_vs_single"] >= max(e["speedup_vs_single"], e["term_range_mode"]["speedup_vs_single"]) - 0.01


def test_bench_is_one_function_per_leg():
    """main() only orchestrates; every leg is a method of the shared context or a module-level function"""
    import inspect
    for name in ("setup", "leg_prove", "leg_unprepared", "leg_one_shot", "leg_batch_c_abi", "leg_ntt", "leg_msm", "leg_msm_strong", "leg_prove_strong",
                 "north_star", "rooflines_msm", "rooflines_prove", "run_leg", "rank0_leg"):
        assert callable(getattr(bench.Bench, name)), name
    for name in ("config2_leg", "config5_leg", "criterion_leg", "sensitivity_legs", "dense_circuit", "cpu_baseline_leg", "protocol_shaped_msm", "in_process_legs"):
        assert callable(getattr(bench, name)), name
    assert len(inspect.getsource(bench.main).splitlines()) < 160            # (it was 700 lines)
    assert bench.scalar_muls_executed(1 << 18, 2, True) in (44 * (1 << 18) + 41, 45 * (1 << 18) + 42)      # (with / without C over the symmetric sums)
    assert bench.scalar_muls_executed(1 << 14, 2, True) == 45 * (1 << 14) + 42


def test_dense_circuit_is_satisfied(ref):
    """the dense-weights sensitivity proves a circuit that HOLDS: aO = aL o aR and every linear constraint wL aL + wR aR + wO aO = cs
    (test/Test/Reference.hs:138,164-169), with no repeated value in the sampled rows"""
    from sonic_amd.workload import rand_fr_array
    n, Q = 64, 3
    c = bench.dense_circuit(rand_fr_array, 7, n, Q)
    val = lambda a: [int.from_bytes(a[i].tobytes(), "little") for i in range(a.shape[0])]      # noqa: E731
    aL, aR, aO = val(c["aL"]), val(c["aR"]), val(c["aO"])
    assert all(o == a * b % ref.R for a, b, o in zip(aL, aR, aO))
    wL, wR, wO, cs = val(c["wL"]), val(c["wR"]), val(c["wO"]), val(c["cs"])
    for q in range(Q):
        lhs = sum(wL[q * n + i] * aL[i] + wR[q * n + i] * aR[i] + wO[q * n + i] * aO[i] for i in range(n)) % ref.R
        assert lhs == cs[q]
        assert len(set(wL[q * n:(q + 1) * n])) == n


def test_effective_cores_respects_quota(monkeypatch, tmp_path):
    n = bench.effective_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
