"""CPU tests of bench.py's host logic: the per-launch-kind byte table adds up to SURVEY 8d's algorithmic bytes per
token, the PMC parser applies the gfx950 correction, and `--gpus N` without a torchrun environment tries to start its
own ranks (and says so loudly when the devices are not there)."""
import os
import subprocess
import sys

from conftest import ROOT
from nano_amd import modelfile as mf

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_kernel_bytes_add_up_to_survey_figures():
    for name, quant, gs in (("qwen3-0.6b", "q80", 64), ("qwen3-0.6b", "q4k", 0), ("nano-168m", "f32", 0), ("qwen3-4b", "q80", 64)):
        spec = mf.preset(name, quant, group_size=gs)
        kb = bench.kernel_bytes(spec, 1, 255)
        weights = sum(kb[k] for k in ("qkv_gemv", "wo_gemv", "w1w3_gemv", "w2_gemv", "classifier_gemv"))
        assert weights == spec.algorithmic_bytes_per_token()
        assert kb["attention"] == 8 * spec.n_layer * spec.kv_dim * 256          # KV_bytes(pos) = 8 L kv_dim (pos+1), SURVEY 8d
    assert bench.kernel_bytes(mf.preset("qwen3-0.6b", "q80", 64), 1, 0)["classifier_gemv"] == 165306368


def test_pmc_csv_parser_applies_the_gfx950_correction(tmp_path):
    p = tmp_path / "c.csv"
    p.write_text("Kernel_Name,Counter_Name,Counter_Value\n"
                 "\"void nano::gemv_q80_stream_kernel<1, 64, 1, 1>(x)\",FETCH_SIZE,80000.0\n"
                 "\"void nano::gemv_q80_stream_kernel<1, 64, 1, 1>(x)\",FETCH_SIZE,82000.0\n"
                 "\"other\",FETCH_SIZE,5.0\n")
    assert bench.parse_pmc_csv(str(p)) == int(81000.0 * 1024 * 2)


def test_gpus_flag_self_launches_or_fails_loudly():
    from nano_amd import binding as nb
    if nb.device_count() >= 2:
        import pytest
        pytest.skip("two devices visible: the real launch is the driver's job")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "--gpus 2 but only" in r.stderr, r.stderr[-500:]
    assert r.stdout.strip() == ""                      # no JSON line claiming a result
