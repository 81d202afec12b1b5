"""bench.py prints ONE JSON line with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the `roofline` and
`cpu_baseline` objects; checked here on a small corpus so that a contract break shows up as a test failure."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--docs", "60000",
           "--chunks", "250000", "--terms", "50000", "--latency-queries", "2", "--cpu-queries", "2", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_has_the_contract_fields():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    d = _run()
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "queries/sec" and d["value"] > 0 and d["ms_per_step"] > 0 and d["data"] == "synthetic"
    assert abs(d["value"] - d["config"]["queries_per_step"] * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["config"]["queries_per_step"] == 256
    assert "model" not in d["config"] and d["config"]["workload"].startswith("hybrid: 60000 docs")
    assert "f64" in d["dtype"] and d["outputs_sane"] is True and d["bm25_parity_vs_cpu"] is True
    r = d["roofline"]
    # 256 queries per step: ONE pass of the 256-query streaming kernel over the f32 rows (csrc/msr_gemm_f32.hip) per step
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "gemm_stream256_kernel<emit>"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r and r["launches"] == 3   # 1 pass x 3 steps
    assert r["bm25_taat"]["launches"] == 3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "queries/sec" and c["sample"]
    for leg in ("port", "vectorised", "literal"):          # three legs, p50 per query, cores stated
        assert c[leg]["p50_ms"] > 0 and c[leg]["value"] > 0 and c[leg]["cores"] >= 1 and c[leg]["queries"] == 2, leg
    assert c["literal"]["cores"] == 1 and c["p50_ms"] == c["port"]["p50_ms"]
    # dense parity against the CPU restatement is reported with every benchmark line
    p = d["dense_parity_vs_cpu"]
    assert p["queries"] == 2 and p["k"] == 100 and p["within_1e-5"] is True and p["max_abs_score_diff"] <= 1e-5
    assert p["top_k_doc_sets_equal_up_to_boundary_near_ties"] is True and p["same_doc_at_same_rank"] > 0.95
    assert abs(d["vs_cpu_baseline"] - d["value"] / c["value"]) <= 1e-9 * d["vs_cpu_baseline"]
    e = d["variant_with_encoder"]                             # query path from token ids (search_api.py:69-152 in one number)
    assert e["value"] > 0 and e["encoder_ms_per_batch"] > 0 and e["outputs_sane"] is True and e["queries_per_step"] == 256
    v = d["variant_bf16_candidates"]
    assert v["value"] > 0 and v["top100_equals_default_path_within_2e-6"] is True


def test_exact_f32_line_reports_the_matrix_core_roofline():
    d = _run("--scan-variant", "2", "--no-cpu-baseline", "--no-variants")
    assert d["dtype"].startswith("f32 (dense cosine)") and "cpu_baseline" not in d and "variant_bf16_candidates" not in d
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and 0 < r["frac"] < 1
