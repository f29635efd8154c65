"""The N > 1 path on CPU: two gloo ranks shard a frame in interleaved row blocks, gather to rank 0 and
re-assemble it, exactly as bench.py does with RCCL on GPUs (the render itself is replaced by a pattern that
encodes each pixel's linear index, so any mis-routed row is visible)."""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

from par_raytracer_amd import capi, sharding

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from par_raytracer_amd import sharding
height, width, block_rows = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rows = sharding.shard_row_list(height, block_rows, rank, world)
max_rows = sharding.max_shard_rows(height, block_rows, world)
shard = torch.zeros((max_rows, width, 4), dtype=torch.float32)
# "render": channel 0 = linear pixel index, channel 1 = rank, channel 3 = 1
for i, y in enumerate(rows):
    shard[i, :, 0] = torch.arange(y * width, (y + 1) * width, dtype=torch.float32)
    shard[i, :, 1] = rank
    shard[i, :, 3] = 1.0
gl = [torch.empty_like(shard) for _ in range(world)] if rank == 0 else None
dist.gather(shard, gl, dst=0)
rays = torch.tensor([float(len(rows) * width)], dtype=torch.float64)
dist.all_reduce(rays, op=dist.ReduceOp.SUM)
if rank == 0:
    idx = torch.from_numpy(sharding.row_index(height, block_rows, world))
    frame = sharding.assemble(torch.cat(gl, dim=0), idx, height)
    expect = torch.arange(height * width, dtype=torch.float32).reshape(height, width)
    assert torch.equal(frame[:, :, 0], expect), "rows mis-assembled"
    assert torch.all(frame[:, :, 3] == 1.0)
    owner = (torch.arange(height) // block_rows) % world
    assert torch.equal(frame[:, 0, 1].to(torch.int64), owner)
    assert int(rays.item()) == height * width
    print("OK")
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("height,width,block_rows,world", [(1080, 16, 8, 2), (37, 5, 8, 2), (64, 3, 5, 3)])
def test_gather_and_reassemble(height, width, block_rows, world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(height), str(width), str(block_rows)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    assert b"OK" in outs[0][0]


def test_python_sharding_matches_the_c_abi():
    lib = capi.hip_lib()
    for h, br, n in ((1080, 8, 8), (2160, 8, 4), (17, 8, 2), (5, 8, 8), (100, 7, 3)):
        for r in range(n):
            assert sharding.shard_rows(h, br, r, n) == lib.prt_shard_rows(h, br, r, n)
        idx = sharding.row_index(h, br, n)
        real = idx[idx < h]
        assert sorted(real.tolist()) == list(range(h))      # every image row exactly once


def test_bench_launches_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` with no launcher around it starts its ranks as child processes.  Without a GPU every
    rank stops with the "needs a GPU" message, and the launcher must hand that failure on as a non-zero exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""        # also on a GPU box: this test is about the launcher, not the render
    env["ROCR_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--workload", "tiny"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = p.stderr.decode()
    assert p.returncode != 0, err[-2000:]
    assert err.count("needs a GPU") >= 1, err[-2000:]
    assert "exited with code" in err


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_rehearsal():
    """The N > 1 path end to end as the driver starts it (`python bench.py --gpus 2`, no torchrun): two ranks share GPU 0,
    gloo stands in for RCCL.  The assembled frame must be bit-identical to the single-GPU render."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
                        "--steps", "3", "--warmup", "1", "--workload", "tiny"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["multi_gpu_check"]["bit_identical_to_single_gpu"] is True
