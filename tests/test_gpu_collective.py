"""RCCL at world size 1: the N > 1 machinery of bench.py with ONE rank, on the one GPU a test box has.

The driver's scaling run needs an 8-GPU node that has not been available to any round, so until round 4 not one call of
`torch.distributed` with the "nccl" (= RCCL) backend had ever executed.  `bench.py --gpus 1 --force-collective` initialises
the process group on the device, creates the render contexts with the CU reservation the multi-GPU path uses
(PRT_RESERVE_CUS=8: CU-masked, i.e. blocking, render streams), keeps two frames in flight, and sends every frame's shard
through dist.gather to rank 0 - itself - on the side stream, then assembles it: library load, communicator creation, the
process group's stream ordering against the persistent render kernels and the assembly are the real thing; only the xGMI
transfer is missing.  Replaces the MPI_Gather of /root/reference main.cpp:345-347.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_collective_path_runs_with_rccl_at_world_size_one():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--backend", "nccl",
                          "--workload", "tiny", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-other-workloads"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1
    c = line["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["forced_at_world_size_1"] and c["reserved_cus"] == 8
    assert c["frames_in_flight"] == 2 and c["gathers_timed"] == 6 and c["gather_and_assemble_ms_mean"] > 0.0
    # the frame that went through gather + assemble is the frame prt_render gives, bit for bit
    assert line["multi_gpu_check"]["bit_identical_to_single_gpu"] is True and line["multi_gpu_check"]["max_abs_diff"] == 0.0
    assert line["value"] > 0 and line["config"]["frames_in_flight"] == 2
