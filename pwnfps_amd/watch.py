"""Deadlines and progress marks for hosts that drive several rank processes (bench.py --gpus N, pwnfps_amd/dist.py's tests).

The reference's frame loop has no peers to lose (main.c:93-109) and its error model is print-and-return
(level.h:35-37,110-115).  A row tiling adds failures of a new kind -- a rank that never arrives, a communicator that does
not come up, a peer that dies inside an exchange -- and each of them, left alone, is a hang: every other rank waits
inside a collective until somebody's patience (the launcher's, the driver's) runs out, and what was lost is the
diagnosis.  Two pieces, pure host code, no GPU:

  Board   every rank writes where it is (`mark("tiled_init")`) into the control plane's key-value store.  Writing and
          reading are NOT collective: a rank that hangs cannot keep the others from reading how far everybody got.
  Watch   a deadline per phase.  When one passes, rank 0 prints ONE JSON line -- "value": null, "incomplete": true,
          the stage every rank reached, how long ago, the last error it recorded -- and every rank leaves with a
          non-zero status (os._exit: no interpreter shutdown, no re-exec of a process that touched the GPU).  The
          other ranks leave a few seconds after rank 0, so that a launcher which tears the job down when the first
          rank exits does not take rank 0 with it before it has printed.  SIGTERM from such a launcher (a peer
          crashed) ends the same way: the signal is picked up by a thread of its own, whatever the main thread is
          stuck in.
"""
import json
import os
import signal
import socket
import sys
import threading
import time


class Board:
    """Progress marks of `world` ranks in a torch.distributed store (TCPStore / the process group's own)."""

    def __init__(self, rank, world, store=None, prefix="pwn/board"):
        self.rank, self.world, self.store, self.prefix = rank, world, store, prefix
        self.local = {"stage": "start", "t": time.time(), "error": None}
        self.t0 = time.time()

    def _key(self, r):
        return "%s/%d" % (self.prefix, r)

    def mark(self, stage, error=None, **extra):
        """This rank is now in `stage`; never raises (a store that is gone must not take the run with it)."""
        self.local = dict(extra, stage=stage, t=time.time(), error=error, host=socket.gethostname(), pid=os.getpid())
        if self.store is not None:
            try:
                self.store.set(self._key(self.rank), json.dumps(self.local))
            except Exception:  # noqa: BLE001
                pass

    def note_error(self, error):
        """the same mark with an error text added (the stage and its time stay)"""
        self.local = dict(self.local, error=error)
        if self.store is not None:
            try:
                self.store.set(self._key(self.rank), json.dumps(self.local))
            except Exception:  # noqa: BLE001
                pass

    def snapshot(self):
        """What every rank last said: [{rank, stage, seconds_ago, error, ...} or {rank, stage: None}]"""
        now = time.time()
        out = []
        for r in range(self.world):
            rec = None
            if r == self.rank:
                rec = self.local
            elif self.store is not None:
                try:
                    if self.store.check([self._key(r)]):
                        rec = json.loads(self.store.get(self._key(r)).decode())
                except Exception:  # noqa: BLE001
                    rec = None
            if rec is None:
                out.append({"rank": r, "stage": None, "note": "no mark from this rank (it never got as far as the control plane, or the store is gone)"})
            else:
                d = {k: v for k, v in rec.items() if k != "t"}
                d.update(rank=r, seconds_ago=round(now - rec["t"], 2))
                out.append(d)
        return out


class Watch:
    """Deadlines for the phases of a multi-rank run.

    make_line(reason, stages) -> dict is called on rank 0 when a deadline passes (or the launcher sends SIGTERM) and
    returns the JSON line to print; it must not touch the GPU or any collective."""

    def __init__(self, board, make_line, grace=4.0, exit_code=3, out=None):
        self.board, self.make_line, self.grace, self.exit_code = board, make_line, grace, exit_code
        self.out = out or sys.stdout
        self.timer = None
        self.phase = None
        self.lock = threading.Lock()
        self.fired = False
        self._sig_r = self._sig_w = None

    # -- deadlines ---------------------------------------------------------------------------------------------------
    def arm(self, seconds, phase):
        """(re)start the clock: `phase` has `seconds` from now"""
        self.disarm()
        if seconds is None or seconds <= 0:
            return
        self.phase = phase
        self.timer = threading.Timer(seconds, self._expired, args=(seconds, phase))
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None
        self.phase = None

    def _expired(self, seconds, phase):
        self.bail("deadline: '%s' did not finish within %.0f s" % (phase, seconds))

    def bail(self, reason):
        """print the diagnostic line (rank 0) and leave, every rank non-zero; never returns"""
        with self.lock:
            if self.fired:
                time.sleep(60)
                os._exit(self.exit_code)
            self.fired = True
        try:
            if self.board.rank == 0:
                stages = self.board.snapshot()
                try:
                    text = json.dumps(self.make_line(reason, stages))
                except Exception as e:  # noqa: BLE001 -- (the main thread may be changing what the line is made of): the bare facts then
                    text = json.dumps({"value": None, "incomplete": True, "error": reason, "line_error": repr(e), "stage_reached": stages})
                self.out.write(text + "\n")
                self.out.flush()
            else:
                # rank 0 first: a launcher that ends the job at the first exit must find its line printed
                self.board.note_error("left: " + reason)
                time.sleep(self.grace)
        finally:
            os._exit(self.exit_code)

    # -- the launcher's SIGTERM ----------------------------------------------------------------------------------------
    def catch_sigterm(self):
        """SIGTERM (torch.distributed.run sends it to every rank when one rank has failed) -> the diagnostic line.  The
        handler only writes a byte to a pipe (signal.set_wakeup_fd: done in C, at once, whatever the main thread is
        blocked in); a thread reads the pipe and bails."""
        if threading.current_thread() is not threading.main_thread():
            return
        self._sig_r, self._sig_w = os.pipe()
        os.set_blocking(self._sig_w, False)
        signal.set_wakeup_fd(self._sig_w, warn_on_full_buffer=False)
        signal.signal(signal.SIGTERM, lambda *_: None)         # (a Python-level handler must exist for the wake-up to be written)

        def reader():
            while True:
                try:
                    b = os.read(self._sig_r, 1)
                except OSError:
                    return
                if b and b[0] == signal.SIGTERM:
                    self.bail("SIGTERM from the launcher while in '%s' (it ends the job when a rank has exited: see the other ranks' marks)"
                              % (self.phase or self.board.local.get("stage")))
        t = threading.Thread(target=reader, daemon=True)
        t.start()


def default_store():
    """the key-value store behind torch.distributed's default process group (rank 0 serves it over TCP), or None"""
    try:
        import torch.distributed as dist
        from torch.distributed import distributed_c10d as c10d
        if dist.is_initialized():
            return c10d._get_default_store()
    except Exception:  # noqa: BLE001
        pass
    return None
