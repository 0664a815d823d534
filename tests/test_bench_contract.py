"""Host-side pieces of bench.py that do not need a GPU: the command-line contract of the driver
(`python bench.py --gpus N --steps K --warmup W`, defaults = N 1 and a K / W that finish within minutes), the self-spawn of
`--gpus N > 1` (one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1) and the committed profile records the
bench line reads its `roofline.traffic` from."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_and_flags(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.workload, a.model, a.cfg_batch, a.fp8) == (1, "dit", "XL", 3, False)
    assert 1 <= a.warmup <= a.steps <= 20          # the default run must finish within minutes
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 2)
    assert b.SAMPLING_STEPS == 30 and b.MFMA_BF16_PEAK_TFLOPS == 2500.0


def test_self_spawn_uses_one_rank_per_gpu_on_localhost(monkeypatch):
    b = _bench()
    seen = {}
    import subprocess

    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert b._self_spawn(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and os.path.basename(cmd[-5]) == "bench.py"


def test_traffic_records_the_bench_line_cites_exist():
    recs = json.load(open(os.path.join(ROOT, "profiles", "kernel_traffic.json")))
    conv = [r for r in recs if r["kernel"].startswith("conv256")]
    assert conv and all(os.path.exists(os.path.join(ROOT, r["source"])) for r in recs)
    assert conv[-1]["hbm_bytes_per_step"] == int(conv[-1]["fetch_size_kb"] * 1024 * 2 + conv[-1]["write_size_kb"] * 1024)
    attn = json.load(open(os.path.join(ROOT, "profiles", "attn_traffic.json")))
    assert any(r["kernel"] == "attn_asm72_kernel" and r["shape"] == [3, 16, 16896, 72] and
               os.path.exists(os.path.join(ROOT, r["source"])) for r in attn)
