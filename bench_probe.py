"""Sandboxed check of the direct transport: `python bench_probe.py`, started by bench_tiled.direct_transport_sandbox() as a CHILD
of every rank of a multi-GPU run before the run itself touches the transport.

The direct transport stores into other processes' (other GPUs') memory from inside a kernel.  If peer mapping is not what it looks
like on a machine, the symptom is not an error code but a GPU memory fault -- which aborts the process.  The children take that
risk: they rendezvous among themselves (gloo, a port of their own), run a few iterations of the same workload on the same tile
grid with the direct transport and compare every tile bit for bit with the single-GPU solver (bench_tiled.direct_transport_precheck).
Exit code 0 on every rank = the transport works here; anything else (a fault, a missed deadline, a mismatch, a child that
never came up) and the parents run the whole leg on RCCL.  Reads SOBFU_PROBE_ARGS (JSON) and the launcher's RANK / WORLD_SIZE /
LOCAL_RANK."""
from __future__ import annotations

import datetime
import json
import os
import sys


class _Ranks:
    def __init__(self, torch, dist, rank, world):
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world

    def _reduce(self, values, op):
        t = self.torch.tensor(list(values), dtype=self.torch.float64)
        self.dist.all_reduce(t, op=op)
        return [float(v) for v in t.tolist()]

    def max(self, values):
        return self._reduce(values, self.dist.ReduceOp.MAX)

    def min(self, values):
        return self._reduce(values, self.dist.ReduceOp.MIN)

    def barrier(self):
        self.dist.barrier()


def main() -> int:
    import numpy as np
    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench_tiled

    a = json.loads(os.environ["SOBFU_PROBE_ARGS"])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    share = os.environ.get("SOBFU_BENCH_SHARE_GPU") == "1"
    torch.cuda.set_device(0 if share else int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("gloo", init_method=f"tcp://{a['addr']}:{a['port']}", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=int(a.get("timeout", 90))))
    if os.environ.get("SOBFU_PROBE_TEST_ABORT") == str(rank):  # tests: this rank's child dies the way a GPU fault would kill it
        os.abort()
    vs = np.array(a["vs"], np.float32)
    P = dict(dims=tuple(a["dims"]), vs=vs, trunc=np.float32(a["trunc"]), eta=np.float32(a["eta"]))
    why = bench_tiled.direct_transport_precheck(P, _Ranks(torch, dist, rank, world), a["kw"], tuple(a["grid"]), iters=int(a.get("iters", 4)))
    if why is not None:
        print(f"direct transport probe, rank {rank}: {why}", file=sys.stderr, flush=True)
    dist.destroy_process_group()
    return 0 if why is None else 3


if __name__ == "__main__":
    sys.exit(main())
