"""One rank of the CPU tests of pwnfps_amd/watch.py (tests/test_watch.py): a stand-in for bench.py's bring-up with the same
Board / Watch objects over a real gloo control plane.
    python tests/watch_rank.py RANK WORLD PORT SCENARIO
SCENARIO  die1   rank 1 leaves (exit 9) when it reaches the stage "tiled_init"; the others sit in a call that never returns
                 (what ncclCommInitRank does when a peer is missing)
          term   nobody dies; every rank sits in that call until the test sends rank 0 a SIGTERM
          ok     every rank gets through; rank 0 prints a normal line"""
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, scenario = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch.distributed as dist
    from pwnfps_amd import watch
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=60))
    board = watch.Board(rank, world, watch.default_store())

    def make_line(reason, stages):
        return {"metric": "test", "value": None, "incomplete": True, "error": reason, "stage_reached": stages}
    w = watch.Watch(board, make_line, grace=1.5)
    w.catch_sigterm()
    board.mark("start")
    w.arm(3.0, "bring-up")
    board.mark("preflight")
    dist.barrier()
    board.mark("tiled_init", transport="fake")
    if scenario == "die1" and rank == 1:
        os._exit(9)
    if scenario in ("die1", "term"):
        print("waiting", flush=True)
        time.sleep(1000)                      # the call that waits for a peer that is not coming
    board.mark("headline")
    w.disarm()
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "test", "value": 1.0, "stage_reached": board.snapshot()}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
