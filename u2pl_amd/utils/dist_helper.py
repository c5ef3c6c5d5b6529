"""Process-group bootstrap with the reference's signature (u2pl/utils/dist_helper.py:13-46): SLURM or torchrun
environment, one process per GPU; backend "nccl" is RCCL over xGMI on ROCm."""
import os
import subprocess

import torch
import torch.distributed as dist


def setup_distributed(backend="nccl", port=None):
    num_gpus = max(torch.cuda.device_count(), 1)
    if "SLURM_JOB_ID" in os.environ:
        rank = int(os.environ["SLURM_PROCID"])
        world_size = int(os.environ["SLURM_NTASKS"])
        addr = subprocess.getoutput("scontrol show hostname {} | head -n1".format(os.environ["SLURM_NODELIST"]))
        if port is not None:
            os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("MASTER_PORT", "10685")
        os.environ.setdefault("MASTER_ADDR", addr)
        os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"], os.environ["RANK"] = str(world_size), str(rank % num_gpus), str(rank)
    else:
        rank, world_size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        if port is not None:
            os.environ.setdefault("MASTER_PORT", str(port))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    if torch.cuda.is_available():
        torch.cuda.set_device(rank % num_gpus)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, world_size=world_size, rank=rank)
    return rank, world_size
