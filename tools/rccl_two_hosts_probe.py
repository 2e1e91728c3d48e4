"""Does RCCL run with TWO ranks on a box that shows ONE GPU?

RCCL refuses two ranks of one host on one device ("Duplicate GPU detected": equal host hash + equal bus id). The host
hash can be overridden per process (NCCL_HOSTID): two ranks that claim different hosts are peers over the NET transport
(sockets on the loopback interface, staged through host memory) and may both sit on cuda:0. That is not xGMI, but it is
RCCL with N = 2: ncclCommInitRank / ncclAllGather / ncclReduce / ncclAllReduce of csrc/collectives.hip, the "nccl"
branches of mi355q/distributed.py and torch's own ProcessGroupNCCL, all with a real peer.

  python tools/rccl_two_hosts_probe.py            # spawns the two ranks, prints one JSON line per rank
"""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def two_hosts_env(rank: int, world: int, port: int) -> dict:
  return dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
              NCCL_HOSTID=f"mi355q-one-gpu-rank-{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET="Socket",
              NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
              NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))


def worker() -> None:
  for p in (os.path.join(ROOT, "ai-edge-quantizer_amd"), ROOT):
    if p not in sys.path:
      sys.path.insert(0, p)
  rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
  import ctypes
  import torch
  import torch.distributed as dist
  torch.cuda.set_device(0)
  t0 = time.perf_counter()
  dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
  out = {"rank": rank, "init_s": round(time.perf_counter() - t0, 2)}
  t = torch.full((1 << 20,), float(rank + 1), device="cuda")
  dist.all_reduce(t)
  torch.cuda.synchronize()
  out["torch_all_reduce"] = float(t[0].item())
  import __graft_entry__ as g
  g.build()
  from mi355q import _ffi, distributed as D, runtime as rt
  L = _ffi.lib()
  comm = D.rccl_comm()
  nr, rk = ctypes.c_int32(-1), ctypes.c_int32(-1)
  _ffi.check(L.mi355q_comm_info(comm, ctypes.byref(nr), ctypes.byref(rk)))
  out["comm_info"] = [nr.value, rk.value]
  loc = torch.full((4096,), float(rank), device="cuda")
  allv = torch.empty((world * 4096,), device="cuda")
  _ffi.check(L.mi355q_allgather_minmax(comm, rt.ptr(loc), 4096, rt.ptr(allv), rt.stream_ptr()))
  torch.cuda.synchronize()
  out["allgather"] = allv.view(world, 4096)[:, 0].tolist()
  big = torch.full((4 << 20,), float(rank + 1), device="cuda")
  _ffi.check(L.mi355q_allreduce_sum_f32(comm, rt.ptr(big), big.numel(), rt.stream_ptr()))
  torch.cuda.synchronize()
  out["allreduce_16MiB"] = [float(big[0].item()), float(big[-1].item())]
  t1 = time.perf_counter()
  for _ in range(5):
    _ffi.check(L.mi355q_allreduce_sum_f32(comm, rt.ptr(big), big.numel(), rt.stream_ptr()))
  torch.cuda.synchronize()
  out["allreduce_16MiB_ms"] = round((time.perf_counter() - t1) / 5 * 1e3, 2)
  dist.barrier()
  D.destroy_rccl_comms()
  dist.destroy_process_group()
  print(json.dumps(out), flush=True)


def main() -> int:
  if os.environ.get("MI355Q_PROBE_WORKER") == "1":
    worker()
    return 0
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  procs = []
  for rank in range(2):
    env = dict(os.environ, MI355Q_PROBE_WORKER="1", **two_hosts_env(rank, 2, port))
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
  deadline = time.time() + float(os.environ.get("MI355Q_PROBE_SECONDS", "240"))
  rc = 0
  for p in procs:
    try:
      rc |= p.wait(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
      rc |= 124
  for p in procs:
    if p.poll() is None:
      p.kill()
  print(json.dumps({"two_ranks_on_one_gpu_rc": rc}), flush=True)
  return rc


if __name__ == "__main__":
  sys.exit(main())
