"""CPU rehearsal of bench.py's guarded all-gather leg (tests/test_bench_launch.py): a gloo rank that builds a stand-in
compute-only line and runs ``bench.guarded_leg`` around a leg that, on the rank named in the environment, hangs, dies
or raises -- the three ways an exchange goes wrong.  Launched as N processes with RANK / WORLD_SIZE / MASTER_* set, by
bench.spawn_ranks-style code in the test or by ``python -m torch.distributed.run``."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch                    # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench                    # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    timeout = float(os.environ.get("GUARD_TIMEOUT", "6"))
    dist.init_process_group("gloo")
    line = {"value": 1.0, "n_gpus": world} if rank == 0 else None
    if rank == 0:
        time.sleep(float(os.environ.get("GUARD_RANK0_EXTRA", "0")))     # rank 0's fill probe + oracle check

    def leg():
        if os.environ.get("GUARD_HANG", "") == str(rank):
            time.sleep(timeout + 120.0)
        if os.environ.get("GUARD_CRASH", "") == str(rank):
            os._exit(7)
        if os.environ.get("GUARD_RAISE", "") == str(rank):
            raise RuntimeError("exchange refused")
        t = torch.ones(4)
        dist.all_reduce(t)
        dist.barrier()
        return {"sum": float(t[0])}

    failed = bench.guarded_leg(line, rank, timeout, dist.barrier, leg)
    if failed:
        os._exit(3)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
