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


def test_effective_cores_respects_quota(monkeypatch, tmp_path):
    n = bench.effective_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
