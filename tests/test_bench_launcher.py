"""bench.py --gpus N: a plain `python bench.py --gpus N` must start N ranks itself (VERDICT r03 #1:
the driver's command line carries no launcher)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
  env.update(kw)
  return env


def test_launch_command_names_every_rank():
  sys.path.insert(0, ROOT)
  import bench
  cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], 12345)
  assert cmd[1:3] == ["-m", "torch.distributed.run"]
  assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
  assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
  assert cmd[-5:] == [BENCH, "--gpus", "4", "--steps", "3"]


def test_plain_python_gpus_2_starts_two_ranks_here():
  """No GPU in this container: both ranks must come up, say which rank of how many they are,
  and refuse to run (the product path has no CPU fallback)."""
  import torch
  if torch.cuda.is_available():
    pytest.skip("CPU-side check of the launcher")
  r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(MI355Q_BENCH_BACKEND="gloo"),
                     capture_output=True, text=True, timeout=300)
  assert r.returncode != 0
  assert "rank 0 of 2" in r.stderr and "rank 1 of 2" in r.stderr, r.stderr[-2000:]


def test_rccl_launch_refuses_more_ranks_than_gpus():
  import torch
  have = torch.cuda.device_count() if torch.cuda.is_available() else 0
  r = subprocess.run([sys.executable, BENCH, "--gpus", str(have + 1 if have else 2)], env=_env(), capture_output=True, text=True, timeout=300)
  assert r.returncode != 0 and "GPU(s) visible" in r.stderr


def test_gpus_flag_must_agree_with_the_launcher():
  r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                     capture_output=True, text=True, timeout=300)
  assert r.returncode != 0 and "must agree" in r.stderr


@pytest.mark.gpu
def test_plain_python_gpus_2_prints_n_gpus_2_on_one_gpu():
  """Two gloo ranks sharing cuda:0 from a plain `python bench.py --gpus 2`: rank 0's line says n_gpus 2
  and carries the sharded configurations' per-rank accounting."""
  r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--cpu-seconds", "0", "--extras", "0"],
                     env=_env(MI355Q_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stderr[-3000:]
  line = json.loads(r.stdout.strip().splitlines()[-1])
  assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "weak"
  assert line["collectives"]["transport"] == "gloo"


@pytest.mark.gpu
def test_plain_python_gpus_2_over_rccl():
  """`python bench.py --gpus 2` over RCCL: one rank per GPU where two are visible, else (MI355Q_BENCH_ONE_GPU_HOSTS=1) the two
  ranks as separate hosts on cuda:0 over RCCL's socket transport. The line must come from an RCCL communicator of two ranks
  whose all-gather check passed -- anything else is exit status 3 with an `error` key."""
  import torch
  extra = {} if torch.cuda.device_count() >= 2 else {"MI355Q_BENCH_ONE_GPU_HOSTS": "1"}
  r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--cpu-seconds", "0", "--extras", "0"],
                     env=_env(**extra), capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stderr[-3000:]
  line = json.loads(r.stdout.strip().splitlines()[-1])
  assert line["n_gpus"] == 2 and "error" not in line
  assert line["collectives"]["transport"] == "rccl via libmi355q"
  assert line["collectives"]["rccl_ranks"] == 2 and line["collectives"]["allgather_correct"] is True
