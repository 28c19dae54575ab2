"""Row-tiled frames across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).
The reference parallelises the same loops with OpenMP over rows
(screen.h:63,77); here rank r owns the contiguous strip of rows

    [r * rows_per, min((r + 1) * rows_per, h))        rows_per = 8-row multiple

The trace pass is independent per pixel.  The blur pass is not strip-local:
its taps reach +-0.002*h*(depth-1) rows (screen.h:86,100-102), unbounded in
depth, so every rank needs the whole pre-blur colour frame.  Per frame:

    1. trace own strip          -> pre[strip], z[strip]            (HIP kernel)
    2. all-gather pre strips    -> pre[all rows]   4 B/pixel       (RCCL, in place)
    3. blur own strip           -> out[strip]                      (HIP kernel)
    4. gather out strips to rank 0                                 (RCCL)

With the blur disabled steps 2-3 vanish.  The collectives are the only data
exchanged; level and sphere tables are uploaded by every rank itself (13 KB).

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


class RowTiledFrame:
    """Frame buffers + choreography for one rank.  Buffers are padded to
    world * rows_per rows so that all strips have equal size for the
    collectives; rows >= h are never written by the kernels."""

    def __init__(self, w, h, backend, device, rank=None, world=None, blur_passes=1, group=None):
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
        for _ in range(self.blur_passes):
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
            self._slot = [dict(pre=self.pre, out=self.out, z=self.z, fin=fin, ag=None, g=None),
                          dict(pre=mk(self.pre), out=mk(self.out), z=mk(self.z),
                               fin=(mk(fin) if fin is not None else None), ag=None, g=None)]
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
        if self.world > 1 and self.blur_passes:
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
        return res if (self.rank == 0) else None

    def to_host(self, t):
        """uint32 numpy view of rows [0,h) of a frame tensor."""
        return t[:self.h].cpu().numpy().view(np.uint32)
