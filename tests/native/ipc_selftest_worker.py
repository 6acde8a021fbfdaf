"""One rank of the peer-store communicator's known-answer run (launched by tests/test_slabs_multiprocess.py under torch.distributed.run, gloo): every rank is its
own OS process on the SAME GPU, maps the others' device windows (fy_comm_create_ipc) and runs fy_comm_selftest -- the grouped two-field neighbour exchange, sum /
max / mixed all-reduces, the all-gather and an exchange on the auxiliary stream beside an all-reduce -- plus a large all-gather and an exchange bigger than a slot
(chunked).  Prints "IPC-SELFTEST OK <rank>"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import conftest  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    prod = conftest.load_product()
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = prod.GlooIpcComm(dist, 0)
    for _ in range(3):                                  # (more than two messages per channel: the slots are reused)
        prod.comm_selftest(comm.handle, 0)
    st = comm.stats()
    assert st["exchanges"] >= 6 and st["allreduces"] >= 12 and st["allgathers"] >= 3, st
    print(f"IPC-SELFTEST OK {rank}", flush=True)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
