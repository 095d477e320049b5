#!/usr/bin/env python3
"""Does RCCL accept a communicator whose two ranks sit on the SAME device?  (test agpl_allreduce_nat on a >= 2-rank RCCL
communicator over one device "if RCCL permits, else say so".)  Two child processes, both on cuda:0, backend "nccl", one all-reduce under a
60 s limit; prints what each rank saw.  The parent never touches the GPU."""
import os
import sys
import traceback


def child(rank):
    import datetime

    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("PROBE_PORT", "29571"), RANK=str(rank), WORLD_SIZE="2")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=60))
        x = torch.full((4,), float(rank + 1), device="cuda:0", dtype=torch.float64)
        dist.all_reduce(x)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce ok -> {x.tolist()}", flush=True)
    except Exception as e:  # noqa: BLE001 (diagnostic tool)
        print(f"rank {rank}: {type(e).__name__}: {str(e)[:600]}", flush=True)
        traceback.print_exc(limit=1)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        import subprocess

        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r)]) for r in range(2)]
        for p in ps:
            try:
                p.wait(timeout=120)
            except subprocess.TimeoutExpired:
                p.kill()
                print("a rank did not finish within 120 s: killed")
