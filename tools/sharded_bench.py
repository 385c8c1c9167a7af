"""Run bench.py's one-Cholesky-over-all-ranks extra on its own: `python tools/sharded_bench.py WORLD N [BACKEND]`.
WORLD ranks are spawned on the GPUs present (rank r on GPU r % device_count); BACKEND gloo moves the collectives through
host memory, which lets two ranks share the one GPU of a test box (a functional run, not a timing)."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, backend, n):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    import bench
    for rep in range(2):
        out = bench.sharded_cholesky_extra(torch, dist, n)
        if rank == 0:
            print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    backend = sys.argv[3] if len(sys.argv) > 3 else "nccl"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(world, port, backend, n), nprocs=world, join=True)
