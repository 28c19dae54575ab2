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

    def to_host(self, t):
        """uint32 numpy view of rows [0,h) of a frame tensor."""
        return t[:self.h].cpu().numpy().view(np.uint32)
