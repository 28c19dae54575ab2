"""Row-tiled frames across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).
The reference parallelises the same loops with OpenMP over rows
(screen.h:63,77); here rank r owns the contiguous strip of rows

    [r * rows_per, min((r + 1) * rows_per, h))        rows_per = 8-row multiple

The trace pass is independent per pixel.  The blur pass is not strip-local:
its taps reach +-0.002*h*(depth-1) rows (screen.h:86,100-102), unbounded in
depth, so every rank needs the whole pre-blur colour frame.  Per frame:

    1. trace own strip          -> pre[strip], z[strip]            (HIP kernel)
    2. exchange pre-blur rows                                      (RCCL)
         "halo" (default): H rows with each neighbour strip, one all-to-all
             with zero-length parts for all other ranks; the blur kernel
             counts taps that land outside [y0-H, y1+H) and, if any rank saw
             one, the frame is repeated with the whole frame gathered
         "allgather": every strip to everyone, 4 B/pixel, in place
    3. blur own strip           -> out[strip]                      (HIP kernel)
    4. gather out strips to rank 0                                 (RCCL)

With the blur disabled steps 2-3 vanish.  Why the halo: xGMI is point to point,
so with 2 GPUs the all-gather and the gather share ONE link and move 2 x 16.6 MB
per 4K frame - more time than tracing the frame on one GPU; with 8, rank 0 takes
in 58 MB per frame.  At 4K the taps of level.txt reach 32 rows; H covers depth 24
(104 rows).  The collectives are the only data exchanged; level and sphere tables
are uploaded by every rank itself (13 KB).

The strip work is delegated to a backend with trace_rows()/blur_rows() on
torch tensors: HipStripBackend (the product: libpwnhip.so on this rank's GPU)
or, in the CPU tests, a checker backend supplied by the test itself.
"""
import numpy as np
import torch
import torch.distributed as dist


def strip_rows(h, world):
    """Rows per strip: h/world rounded up to the 8-row tile height of the trace kernel."""
    per = -(-h // world)
    return -(-per // 8) * 8


def strip_range(h, world, rank):
    per = strip_rows(h, world)
    y0 = min(rank * per, h)
    return y0, min(y0 + per, h)


class HipStripBackend:
    """Strips on this process's GPU through the C ABI (device pointers, torch's
    current stream).  There is no CPU path: constructing it without a usable
    GPU raises."""

    def __init__(self, renderer):
        if not torch.cuda.is_available():
            raise RuntimeError("HipStripBackend needs a GPU (libpwnhip.so has no CPU fallback)")
        self.r = renderer
        self.device = torch.device("cuda", renderer.device)

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def trace_rows(self, cam, sec, y0, y1, pre, z):
        self.r.trace_rows_device(cam, sec, y0, y1, pre.data_ptr(), z.data_ptr(), self._stream())

    def blur_rows(self, y0, y1, pre, z, out):
        self.r.blur_rows_device(y0, y1, pre.data_ptr(), z.data_ptr(), out.data_ptr(), self._stream())

    def blur_rows_bounded(self, y0, y1, pre, z, out, avail_y0, avail_y1, miss):
        """miss: int32 device tensor (1 element) the kernel adds to for taps outside [avail_y0, avail_y1)."""
        self.r.blur_rows_device_bounded(y0, y1, pre.data_ptr(), z.data_ptr(), out.data_ptr(), avail_y0, avail_y1,
                                        miss.data_ptr(), self._stream())


class RowTiledFrame:
    """Frame buffers + choreography for one rank.  Buffers are padded to
    world * rows_per rows so that all strips have equal size for the
    collectives; rows >= h are never written by the kernels."""

    def __init__(self, w, h, backend, device, rank=None, world=None, blur_passes=1, group=None,
                 exchange="halo", halo_depth=24.0):
        self.w, self.h = int(w), int(h)
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world = rank, world
        self.blur_passes = int(blur_passes)
        if self.blur_passes > 0 and (self.w & 3):
            raise ValueError("blur needs a width divisible by 4 (screen.h:88)")
        self.backend = backend
        self.per = strip_rows(self.h, world)
        self.hpad = self.per * world
        self.y0, self.y1 = strip_range(self.h, world, rank)
        kw = dict(device=device)
        # int32 views of the uint32 BGRA pixels (collectives do not care)
        self.pre = torch.zeros((self.hpad, self.w), dtype=torch.int32, **kw)
        self.out = torch.zeros((self.hpad, self.w), dtype=torch.int32, **kw)
        self.z = torch.zeros((self.hpad, self.w), dtype=torch.float32, **kw)
        self.final = torch.zeros((self.hpad, self.w), dtype=torch.int32, **kw) if rank == 0 else None
        self.final_z = None
        # bounded exchange: H rows per neighbour, possible when every strip has at least H rows
        self.halo = 0
        self.halo_misses = 0       # frames (render) / flushes (submit) that had to fall back to the all-gather
        if exchange == "halo" and world > 1 and self.blur_passes == 1:
            H = int(np.ceil(0.002 * self.h * float(halo_depth))) + 1
            shortest = min(strip_range(self.h, world, r)[1] - strip_range(self.h, world, r)[0] for r in range(world))
            if 0 < H <= shortest:
                self.halo = H
        elif exchange not in ("halo", "allgather"):
            raise ValueError("exchange must be 'halo' or 'allgather'")
        if self.halo:
            H = self.halo
            up, dn = rank > 0, rank < world - 1
            self._peers = (up, dn)
            self._in_split = [H if (r == rank - 1 or r == rank + 1) else 0 for r in range(world)]
            self._nrows = H * (int(up) + int(dn))
            self.avail = (self.y0 - H if up else 0, self.y1 + H if dn else self.h)
            self._hx = [self._halo_bufs(kw) for _ in range(2)]

    def _halo_bufs(self, kw):
        return dict(send=torch.zeros((max(self._nrows, 1), self.w), dtype=torch.int32, **kw),
                    recv=torch.zeros((max(self._nrows, 1), self.w), dtype=torch.int32, **kw),
                    miss=torch.zeros(1, dtype=torch.int32, **kw))

    def _halo_start(self, pre, hx, async_op):
        """Send this strip's border rows to the neighbours (ascending rank order: the upper
        neighbour gets my top rows, the lower one my bottom rows); returns the work handle."""
        H, (up, dn) = self.halo, self._peers
        k = 0
        if up:
            hx["send"][0:H].copy_(pre[self.y0:self.y0 + H]); k = H
        if dn:
            hx["send"][k:k + H].copy_(pre[self.y1 - H:self.y1])
        n = self._nrows
        return dist.all_to_all_single(hx["recv"][:n], hx["send"][:n], output_split_sizes=self._in_split,
                                      input_split_sizes=self._in_split, group=self.group, async_op=async_op)

    def _halo_finish(self, pre, hx):
        """Place the received rows: the upper neighbour's bottom rows above my strip, the
        lower neighbour's top rows below it."""
        H, (up, dn) = self.halo, self._peers
        k = 0
        if up:
            pre[self.y0 - H:self.y0].copy_(hx["recv"][0:H]); k = H
        if dn:
            pre[self.y1:self.y1 + H].copy_(hx["recv"][k:k + H])

    def _any_miss(self, miss):
        """Collective: did any rank count a tap outside its halo?  (host sync)"""
        m = miss.clone()
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        return int(m.item()) != 0

    def _strip(self, t, rank=None):
        rank = self.rank if rank is None else rank
        return t[rank * self.per:(rank + 1) * self.per]

    def render(self, cam, sec=0.0, gather_depth=False):
        """One frame.  Returns the device tensor holding the final frame on rank 0
        (rows [0,h) valid), None elsewhere."""
        b = self.backend
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        b.trace_rows(cam, float(sec), self.y0, self.y1, self.pre, self.z)
        cur, other = self.pre, self.out
        done = False
        if self.halo:
            hx = self._hx[0]
            hx["miss"].zero_()
            self._halo_start(cur, hx, False)
            self._halo_finish(cur, hx)
            b.blur_rows_bounded(self.y0, self.y1, cur, self.z, other, self.avail[0], self.avail[1], hx["miss"])
            if self._any_miss(hx["miss"]):
                self.halo_misses += 1          # some tap left the halo: repeat with the whole frame
            else:
                cur, other = other, cur
                done = True
        for _ in range(0 if done else self.blur_passes):
            if self.world > 1:
                # in place: this rank's strip is already at its slot in `cur`
                dist.all_gather_into_tensor(cur, self._strip(cur), group=self.group)
            b.blur_rows(self.y0, self.y1, cur, self.z, other)
            cur, other = other, cur
        if self.world == 1:
            self.final_z = self.z
            return cur
        if self.rank == 0:
            parts = [self._strip(self.final, r) for r in range(self.world)]
            dist.gather(self._strip(cur), parts, dst=0, group=self.group)
            if gather_depth:
                if self.final_z is None:
                    self.final_z = torch.zeros_like(self.z)
                zparts = [self._strip(self.final_z, r) for r in range(self.world)]
                dist.gather(self._strip(self.z), zparts, dst=0, group=self.group)
            return self.final
        dist.gather(self._strip(cur), None, dst=0, group=self.group)
        if gather_depth:
            dist.gather(self._strip(self.z), None, dst=0, group=self.group)
        return None

    # -- frames in flight ------------------------------------------------------
    # With world > 1 a frame is trace -> all-gather -> blur -> gather, and at 4K the
    # two collectives take longer than the kernels.  A renderer presents frames in
    # a stream, so consecutive frames are overlapped: while the collectives of
    # frame i are on the wire (RCCL's own stream), the kernels of frame i+1 run.
    # Every buffer exists twice (slot = frame & 1); hazards:
    #   pre[s]  trace_i writes own strip -> all-gather_i fills the rest -> blur_i reads;
    #           next writer trace_{i+2} is issued after blur_i on the same stream
    #   out[s]  blur_i writes own strip -> gather_i reads; next writer blur_{i+2}
    #           waits for gather_i first
    #           (with the blur off the gather reads pre[s]: trace_{i+2} waits for gather_i)
    #   z[s]    trace_i writes, blur_i reads (same stream)
    #   final[s] (rank 0) gather_i writes; one per slot, so two gathers never share a target
    # All ranks issue the collectives in the same order: AG_0, AG_1, G_0, AG_2, G_1, ...
    def _slots(self):
        if getattr(self, "_slot", None) is None:
            mk = lambda t: torch.zeros_like(t)
            fin = self.final
            self._slot = [dict(pre=self.pre, out=self.out, z=self.z, fin=fin, ag=None, g=None, hx=None),
                          dict(pre=mk(self.pre), out=mk(self.out), z=mk(self.z),
                               fin=(mk(fin) if fin is not None else None), ag=None, g=None, hx=None)]
            if self.halo:
                for i in (0, 1):
                    self._slot[i]["hx"] = self._hx[i]
                self._miss_run = torch.zeros(1, dtype=torch.int32, device=self.pre.device)
            self._last = None      # (cam, sec) of the newest frame, for the fallback in flush()
            self._n = 0            # frames submitted
            self._pending = None   # slot index of the frame traced but not yet blurred
        return self._slot

    def _finish(self, k):
        """blur + gather of the frame in slot k (its all-gather is in flight)."""
        sl = self._slot[k]
        if sl["g"] is not None:            # gather of the frame that used out[k] two submits ago
            sl["g"].wait()
            sl["g"] = None
        cur = sl["pre"]
        if self.blur_passes:
            if sl["ag"] is not None:
                sl["ag"].wait()
                sl["ag"] = None
            if self.halo:
                self._halo_finish(sl["pre"], sl["hx"])
                self.backend.blur_rows_bounded(self.y0, self.y1, sl["pre"], sl["z"], sl["out"],
                                               self.avail[0], self.avail[1], self._miss_run)
            else:
                self.backend.blur_rows(self.y0, self.y1, sl["pre"], sl["z"], sl["out"])
            cur = sl["out"]
        if self.world == 1:
            return cur
        if self.rank == 0:
            parts = [self._strip(sl["fin"], r) for r in range(self.world)]
            sl["g"] = dist.gather(self._strip(cur), parts, dst=0, group=self.group, async_op=True)
        else:
            sl["g"] = dist.gather(self._strip(cur), None, dst=0, group=self.group, async_op=True)
        return sl["fin"]

    def submit(self, cam, sec=0.0):
        """Enqueue one frame (POSTPROC_BLUR 0 or 1).  Returns None; the frame is
        complete after the next submit() or after flush()."""
        if self.blur_passes > 1:
            raise ValueError("frames in flight support blur_passes 0 or 1")
        slots = self._slots()
        k = self._n & 1
        sl = slots[k]
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        if sl["g"] is not None:            # the gather two frames back read this slot's buffers
            sl["g"].wait()
            sl["g"] = None
        self.backend.trace_rows(cam, float(sec), self.y0, self.y1, sl["pre"], sl["z"])
        self._last = (cam.copy(), float(sec))
        if self.halo:
            sl["ag"] = self._halo_start(sl["pre"], sl["hx"], True)
        elif self.world > 1 and self.blur_passes:
            sl["ag"] = dist.all_gather_into_tensor(sl["pre"], self._strip(sl["pre"]), group=self.group, async_op=True)
        if self._pending is not None:
            self._finish(self._pending)
        self._pending = k
        self._n += 1

    def flush(self):
        """Complete everything in flight.  Returns the last frame on rank 0 (device
        tensor, rows [0,h) valid; for world == 1 on that one rank), None elsewhere."""
        slots = self._slots()
        res = None
        if self._pending is not None:
            res = self._finish(self._pending)
            self._pending = None
        for sl in slots:
            if sl["g"] is not None:
                sl["g"].wait()
                sl["g"] = None
        if self.halo and self._last is not None:
            # taps outside the halo anywhere since the last flush?  Then the frames in flight were
            # not exact; the one handed back is rendered again with the whole frame gathered.
            miss = self._any_miss(self._miss_run)
            self._miss_run.zero_()
            if miss:
                self.halo_misses += 1
                H, self.halo = self.halo, 0
                try:
                    res = self.render(*self._last)
                finally:
                    self.halo = H
        return res if (self.rank == 0) else None

    def to_host(self, t):
        """uint32 numpy view of rows [0,h) of a frame tensor."""
        return t[:self.h].cpu().numpy().view(np.uint32)
