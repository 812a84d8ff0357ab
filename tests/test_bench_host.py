"""Host-side accounting of bench.py: the algorithmic-byte formulas against the figures SURVEY.md §8-d states, and the
PMC traffic record bench.py reports as roofline.traffic.  No GPU."""
import json
import os
import sys

import bench
from medical_image_analysis_amd.selective_scan_interface import scan_algorithmic_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scan_algorithmic_bytes_match_survey_figures():
    # cfg#2 fp32 = 77.93 MB (81.07 MB with checkpoints); target B=8 fp32 ~ 809.6 MB, bf16 I/O ~ 405 MB
    assert round(bench.scan_bytes(32, 768, 196, 16, 1, 4) / 1e6, 2) == 77.93
    assert round(bench.scan_bytes(8, 1536, 4096, 16, 1, 4) / 1e6, 1) == 809.6
    assert round(bench.scan_bytes(8, 1536, 4096, 16, 1, 2) / 1e6) == 405
    assert scan_algorithmic_bytes(8, 1536, 4096, 16, 1, 4, True) == bench.scan_bytes(8, 1536, 4096, 16, 1, 4)
    # the reference's checkpoint buffer: (B, D, ceil(L/2048), 2N) fp32 for cfg#2 -> +3.15 MB
    assert round((bench.scan_bytes(32, 768, 196, 16, 1, 4) + 8 * 32 * 768 * 16 * 1) / 1e6, 2) == 81.07
    # backward: 4 reads + 3 writes of (B, D, L) rows, B/C reads in the io dtype, dB/dC in fp32
    bwd = scan_algorithmic_bytes(2, 64, 100, 16, 1, 2, True, backward=True)
    assert bwd == 2 * (7 * 2 * 64 * 100 + 2 * 2 * 16 * 100) + 8 * 2 * 16 * 100 + 2 * 4 * (64 * 16 + 2 * 64)


def test_pmc_traffic_record_feeds_the_roofline_object():
    rec = bench.pmc_traffic("scan_fwd_target")
    assert rec is not None and rec["source"].endswith("_pmc_traffic.json")
    with open(os.path.join(ROOT, "profiles", rec["source"])) as f:
        raw = json.load(f)["scan_fwd_target"]
    # FETCH_SIZE / WRITE_SIZE count KB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md)
    assert rec["bytes"] == int((2 * raw["fetch_kb"] + raw["write_kb"]) * 1024)
    algo = bench.scan_bytes(8, 1536, 4096, 16, 1, 4)
    assert 1.0 <= rec["bytes"] / algo < 1.10        # no wasted re-reads: measured traffic within 10 % of the algorithmic bytes
    assert bench.pmc_traffic("no_such_workload") is None


def test_host_core_count_is_sane():
    n = bench.host_physical_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_decode_cpu_baseline_runs_on_a_tiny_workload(monkeypatch):
    """bench.cpu_baseline_decode: the torch CPU path of the decoder on 2- and 6-layer samples, extrapolated per layer -- here on a
    toy width so that it takes a second; the record must carry a positive rate, the core count and the description of the sample."""
    import torch
    monkeypatch.setitem(bench.DECODE_WORKLOADS, "toy", (128, 64, 96, 8, 4, 4, 16, 8, 2, 2, "toy decoder"))
    threads = torch.get_num_threads()
    try:
        rec = bench.cpu_baseline_decode("toy")
    finally:
        torch.set_num_threads(threads)
    assert rec["unit"] == "tokens/sec" and rec["kind"] == "port" and rec["value"] > 0 and rec["cores"] >= 1
    assert "2- and 6-layer" in rec["sample"] and "8 x" in rec["sample"]


def test_bench_help_renders():
    """`python bench.py --help` (argparse expands `%` in help strings: an unescaped one made the command raise)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "--workload" in r.stdout and "--decode-gemm" in r.stdout, r.stderr[-500:]
