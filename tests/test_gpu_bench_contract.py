"""bench.py's one JSON line on a scaled-down workload: the driver's contract (metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload), the `roofline` and
`cpu_baseline` objects of the tier brief, a leg for every BASELINE config (configs[1] headline, [2] fm, [3] banded, [4]
seed_extend — each with its strong variant where the config is multi-GPU), and every parity flag true."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields_and_green_parity():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "20000",
           "--k1-pairs", "4096", "--genome", "300000", "--fm-big-genome", "400000", "--queries", "20000", "--banded-pairs", "70",
           "--banded-parity-pairs", "70", "--pipeline-reads", "3000", "--pipeline-reads-total", "4001", "--ingest-reads", "3000",
           "--fmd-genome", "200000", "--fmd-reads", "2000"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    d = json.loads(lines[0])
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
        assert isinstance(d[k], typ), k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and r["peak"] == 8000.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # the other BASELINE configs, each with roofline + cpu_baseline + its strong variant
    for leg in ("fm", "fm_big", "seed_extend", "banded", "fmd_smems"):
        assert d[leg]["value"] > 0 and "roofline" in d[leg] and "cpu_baseline" in d[leg], leg
    assert d["fm"]["strong"]["queries_total"] == 20000 and d["banded"]["strong"]["pairs_total"] == 70
    assert d["seed_extend"]["strong"]["reads_total"] == 4001 and d["seed_extend"]["strong"]["gathered_records"] == 4001
    assert d["value_int32"] > 0 and d["int32"]["records_and_ops_equal_int16_run"] is True
    sg = d["semiglobal"]  # north_star: Aligner::local / semiglobal
    assert sg["value"] > 0 and sg["roofline"]["frac"] > 0 and sg["cpu_baseline"]["value"] > 0 and d["parity"]["semiglobal_bit_exact"] is True
    assert d["packed2"]["records_and_ops_equal_byte_run"] is True and d["fm"]["packed2"]["results_equal_byte_run"] is True
    assert d["fm"]["roofline"]["requested_lines_per_launch"] > 0
    flags = {k: v for k, v in d["parity"].items() if isinstance(v, bool)}
    assert len(flags) >= 10 and all(flags.values()), flags
