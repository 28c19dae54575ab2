"""FROZEN (round 5): a second implementation of the row tiling's choreography, kept because the gloo tests
(tests/test_dist_gloo.py) prove the PROTOCOL on CPU with it -- not the product path, and not to be extended.  New host logic
goes into ``pwnfps_amd/csrc/pwn_tiled.cpp`` / ``pwn_group.cpp``, whose C code itself now runs on the CPU in the suite
(tools/sanitize/: stand-in HIP runtime and kernels, TSan / ASan; tests/test_sanitize_host.py).

Row-tiled frames across the GPUs of a node: the choreography of
``pwnfps_amd/csrc/pwn_tiled.cpp`` restated over ``torch.distributed`` point-to-point
operations.

The product path is the C one: ``Renderer.tiled_init / tiled_submit / tiled_wait``
(``include/pwnhip.h``, RCCL inside the library); ``bench.py --gpus N`` and ``host/pwnhost -W N``
use it.  This module is the same state machine in the C code's default order (what goes into which group, which
buffers a frame owns, what every rank decides from the words), with

* the transport = one ``batch_isend_irecv`` per grouped exchange (gloo on CPU, nccl = RCCL on
  GPUs), and
* the strip work delegated to a backend with ``trace_rows / blur_rows / blur_rows_bounded``
  on torch tensors: ``HipStripBackend`` (libpwnhip.so on this rank's GPU) or, in the CPU
  tests, a checker backend supplied by the test,

so that the protocol -- what goes into which group, which buffers a frame owns, when a frame
whose blur taps left the halo is repeated, and that every rank decides the same -- is covered
by world_size 2 / 3 / 8 tests without a GPU (tests/test_dist_gloo.py).

One process per rank; rank r owns rows [cuts[r], cuts[r+1]) -- to begin with the equal split, per =
ceil(h/world) rounded up to 8 (the reference parallelises the same loops with OpenMP over rows, screen.h:63,77);
then the cuts MOVE with what the strips cost (``recut``: every rank's cost word travels with the frame's miss
word, every rank computes the same new cuts from the same numbers, a frame keeps the cuts it was traced with).  The
trace pass needs no exchange.  The blur does: its taps reach 0.002*h*(depth-1) rows
(screen.h:100-102), unbounded in depth.  Per submitted frame f (slot s = f mod NSLOT), in this order -- the C code's
default choreography, PWN_TILED_CHOREO_INSTREAM, where all of it is enqueued on the frame's own compute stream:

    trace strip f -> pre[s], z[s]
    G1(f): the H border rows of strip f to / from the neighbour strips (or, without a halo, every strip to
           everybody)
    blur strip f from pre rows [y0-H, y1+H) -> out; taps outside those rows are counted in the rank's miss
           word of that frame
    G2(f): the FINISHED strip of frame f to the frame's root and its two words to every rank

(In the C code frames alternate between two streams, so that frame f's exchange overlaps the trace of f+1; here
everything is synchronous and only the order matters.  The C code's other choreography, PWN_TILED_CHOREO_SPLIT,
enqueues blur f with submit f+1 and G2(f) with submit f+2 on a third stream: the same groups in another order,
the same frames.)  Frame f is complete on its root when G2(f) is; five frames in flight at
most.  Every rank then holds every rank's miss word of frame f: if one is non-zero ALL ranks repeat
that frame's exchange with whole strips, its blur and its gather before it is delivered, and use
whole strips from then on.

Host sink (``pwn_tiled_host_sink``; here ``host_sink=`` an array of NSLOT frames in memory that every
rank has mapped): there is no gather.  Behind its blur every rank copies its strip into the frame, and
the second half of a group is one word per pair of ranks, sent after the sender's copy; a frame is
delivered, on every rank, when every other rank's word of it has arrived.  (The C code sends that word one submit
late, on the copy's stream in front of the next frame's copy, so that no exchange waits for PCIe; the same groups.)
"""
import numpy as np
import torch
import torch.distributed as dist

TAG_HALO, TAG_STRIP, TAG_GATHER, TAG_MISS = 1, 2, 3, 4
NSLOT = 6          # buffer sets: at most five frames in flight and the one being reused


def strip_rows(h, world):
    """Rows per strip: h/world rounded up to the 8-row multiple the trace kernel tiles by."""
    per = -(-h // world)
    return -(-per // 8) * 8


def strip_range(h, world, rank):
    per = strip_rows(h, world)
    y0 = min(rank * per, h)
    return y0, min(y0 + per, h)


def default_halo(h):
    """rows that cover blur taps up to depth 24 (pwn_tiled_init's default)"""
    return int(0.002 * h * 24.0) + 2


def equal_cuts(h, world):
    per = strip_rows(h, world)
    return [min(r * per, h) for r in range(world)] + [h]


def max_strip_rows(h, world):
    """the tallest strip a rank may be given: one and a half equal strips (pwn_tiled_init)"""
    per = strip_rows(h, world)
    return min(h, (per + per // 2 + 7) // 8 * 8)


def recut(old, cost, world, h, min_rows, max_rows):
    """``recut`` of pwn_tiled.cpp, statement for statement: new cuts from what the strips of a delivered frame cost
    (piecewise-constant cost per row inside a strip; cut k where the running cost reaches k/world of the total, three
    quarters of the way from the old cut, multiples of 8; every strip within [min_rows, max_rows]).  None = leave them."""
    total, top = 0.0, 0.0
    for r in range(world):
        if int(cost[r]) == 0 or old[r + 1] <= old[r]:
            return None
        total += float(cost[r])
        top = max(top, float(cost[r]))
    if top * world < total * 1.02:
        return None
    cut = [0] * (world + 1)
    cut[world] = h
    acc, r = 0.0, 0
    for k in range(1, world):
        want = total * float(k) / float(world)
        while r < world - 1 and acc + float(cost[r]) < want:
            acc += float(cost[r])
            r += 1
        y = float(old[r]) + (want - acc) / float(cost[r]) * float(old[r + 1] - old[r])
        y = float(old[k]) + 0.75 * (y - float(old[k]))
        cut[k] = int(y / 8.0 + 0.5) * 8
    min_rows = max((min_rows + 7) // 8 * 8, 8)
    if min_rows * world > h or max_rows * world < h:
        return None
    for k in range(1, world):
        cut[k] = max(cut[k], cut[k - 1] + min_rows)
        cut[k] = min(cut[k], cut[k - 1] + max_rows)
    for k in range(world - 1, 0, -1):
        if cut[k] > cut[k + 1] - min_rows:
            cut[k] = (cut[k + 1] - min_rows) // 8 * 8
        if cut[k] < cut[k + 1] - max_rows:
            cut[k] = (cut[k + 1] - max_rows + 7) // 8 * 8
    for k in range(1, world + 1):
        if cut[k] - cut[k - 1] < min_rows or cut[k] - cut[k - 1] > max_rows:
            return None
    return cut if cut != list(old) else None


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
        """(returns nothing: the strip forms of the C ABI do not measure cost -- the cuts stay where they are)"""
        self.r.trace_rows_device(cam, sec, y0, y1, pre.data_ptr(), z.data_ptr(), self._stream())

    def blur_rows(self, y0, y1, pre, z, out):
        self.r.blur_rows_device(y0, y1, pre.data_ptr(), z.data_ptr(), out.data_ptr(), self._stream())

    def blur_rows_bounded(self, y0, y1, pre, z, out, avail_y0, avail_y1, miss):
        """miss: int32 device tensor (1 element) the kernel adds to for taps outside [avail_y0, avail_y1)."""
        self.r.blur_rows_device_bounded(y0, y1, pre.data_ptr(), z.data_ptr(), out.data_ptr(), avail_y0, avail_y1,
                                        miss.data_ptr(), self._stream())


class TiledFrames:
    """pwn_tiled_init / _submit / _wait of pwn_tiled.cpp, restated.  Names follow the C code."""

    def __init__(self, w, h, backend, device, rank=None, world=None, blur_passes=1, halo_rows=-1, group=None, host_sink=None,
                 balance_every=None, rotate_root=False):
        self.w, self.h = int(w), int(h)
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world = rank, world
        self.blur_passes = int(blur_passes)
        if self.blur_passes > 1:
            raise ValueError("row tiling supports POSTPROC_BLUR 0 or 1")
        if self.blur_passes > 0 and (self.w & 3):
            raise ValueError("blur needs a width divisible by 4 (screen.h:88)")
        self.backend = backend
        self.per = strip_rows(self.h, world)
        self.cuts = equal_cuts(self.h, world)          # for the next submitted frame
        self.fcuts = [list(self.cuts) for _ in range(NSLOT)]   # ... as used for the frame in that slot
        self.max_rows = max_strip_rows(self.h, world)
        H = default_halo(self.h) if halo_rows < 0 else int(halo_rows)
        shortest = min(self.cuts[r + 1] - self.cuts[r] for r in range(world))
        if world == 1 or self.blur_passes == 0 or H > shortest or H <= 0:
            H = 0
        self.halo = H                       # rows exchanged with each neighbour; 0 = whole strips to everybody
        self.fhalo = [0] * NSLOT            # ... as used for the frame in that slot
        # moving cuts: every `balance_every` delivered frames (pwn_tiled_balance; the C default is 8)
        dflt = 8 if (world > 1 and shortest >= 16 and self.blur_passes == 1) else 0
        self.balance_every = dflt if balance_every is None else (int(balance_every) if dflt else 0)
        self.recuts = 0
        self.last_cost = [0] * world
        # what the re-cut works from: per rank the smallest cost among the delivered frames traced with acc_cuts
        self.acc_cuts, self.acc_cost = None, None
        kw = dict(device=device)
        mk = lambda dt: [torch.zeros((self.h, self.w), dtype=dt, **kw) for _ in range(NSLOT)]   # noqa: E731
        # int32 views of the uint32 BGRA pixels (the transport does not care)
        self.pre, self.out, self.z = mk(torch.int32), mk(torch.int32), mk(torch.float32)
        # pwn_tiled_gather_root: frame f is assembled on rank 0, or on rank f mod world in turn
        self.rotate_root = bool(rotate_root)
        self.froot = [0] * NSLOT            # ... the root of the frame in that slot
        self.fin = mk(torch.int32) if (rank == 0 or self.rotate_root) else [None] * NSLOT
        # the two words a rank says about a frame: [0] taps that left its halo, [1] what its strip cost
        self.missw = [torch.zeros(2, dtype=torch.int32, **kw) for _ in range(NSLOT)]
        self.missv = [torch.zeros(2 * world, dtype=torch.int32, **kw) for _ in range(NSLOT)]
        self.submitted = self.blurred = self.gathered = self.delivered = 0
        self.info = dict(frames=0, frames_redone=0, groups=0, bytes_sent=0, bytes_received=0, bytes_to_host=0)
        # pwn_tiled_host_sink: uint32 [NSLOT, h, w] in memory shared by the ranks (np.memmap of one file ...)
        self.host = host_sink
        if self.host is not None and tuple(self.host.shape) != (NSLOT, self.h, self.w):
            raise ValueError("a host sink holds %d frames of %dx%d" % (NSLOT, self.w, self.h))

    @property
    def y0(self):
        return self.cuts[self.rank]

    @property
    def y1(self):
        return self.cuts[self.rank + 1]

    def set_cuts(self, cuts):
        """pwn_tiled_set_cuts: world + 1 boundaries for the next submitted frames; the same call on every rank"""
        cuts = [int(v) for v in cuts]
        lo = self.halo if self.halo > 0 else 1
        ok = len(cuts) == self.world + 1 and cuts[0] == 0 and cuts[-1] == self.h
        for r in range(self.world):
            rows = cuts[r + 1] - cuts[r] if ok else 0
            ok = ok and lo <= rows <= self.max_rows and rows >= 1 and (r + 1 == self.world or cuts[r + 1] % 8 == 0)
        if not ok:
            raise ValueError("cuts: 0 = c[0] < ... < c[world] = h in multiples of 8, every strip %d..%d rows" % (lo, self.max_rows))
        self.cuts = cuts

    # ---- the transport: a group = operations that progress together -----------------------
    def _begin(self):
        self._ops = []

    def _send(self, t, peer, tag):
        self._ops.append(dist.P2POp(dist.isend, t, peer, group=self.group, tag=tag))
        self.info["bytes_sent"] += t.numel() * 4

    def _recv(self, t, peer, tag):
        self._ops.append(dist.P2POp(dist.irecv, t, peer, group=self.group, tag=tag))
        self.info["bytes_received"] += t.numel() * 4

    def _end(self):
        if self._ops:
            for req in dist.batch_isend_irecv(self._ops):
                req.wait()
        self.info["groups"] += 1

    def _rows_of(self, s, r):
        """strip r of the frame in slot s"""
        return self.fcuts[s][r], self.fcuts[s][r + 1]

    def _add_words(self, s):
        """the two words of every rank to every rank"""
        for r in range(self.world):
            if r == self.rank:
                continue
            self._send(self.missw[s], r, TAG_MISS)
            self._recv(self.missv[s][2 * r:2 * r + 2], r, TAG_MISS)

    # ---- pieces of a group (add_gather / add_allgather of pwn_tiled.cpp) ------------------
    def _add_gather(self, g):
        s = g % NSLOT
        mine = self.out[s] if self.blur_passes else self.pre[s]
        y0, y1 = self._rows_of(s, self.rank)
        if self.host is not None:
            # no strips: the words go out behind this rank's copy to the host
            self._add_words(s)
            return
        root = self.froot[s]
        if self.rank == root:
            for r in range(self.world):
                if r == root:
                    continue
                a, b = self._rows_of(s, r)
                if b > a:
                    self._recv(self.fin[s][a:b], r, TAG_GATHER)
        elif y1 > y0:
            self._send(mine[y0:y1], root, TAG_GATHER)
        self._add_words(s)

    def _add_allgather(self, s):
        y0, y1 = self._rows_of(s, self.rank)
        for r in range(self.world):
            if r == self.rank:
                continue
            a, b = self._rows_of(s, r)
            if y1 > y0:
                self._send(self.pre[s][y0:y1], r, TAG_STRIP)
            if b > a:
                self._recv(self.pre[s][a:b], r, TAG_STRIP)

    def _copy_strip_to_host(self, s):
        y0, y1 = self._rows_of(s, self.rank)
        if y1 > y0:
            src = self.out[s] if self.blur_passes else self.pre[s]
            self.host[s, y0:y1] = src[y0:y1].cpu().numpy().view(np.uint32)
            if hasattr(self.host, "flush"):
                self.host.flush()
            self.info["bytes_to_host"] += (y1 - y0) * self.w * 4

    def _enqueue_blur(self, k):
        s = k % NSLOT
        y0, y1 = self._rows_of(s, self.rank)
        if self.blur_passes:
            dst = self.fin[s] if (self.rank == self.froot[s] and self.host is None) else self.out[s]
            if self.fhalo[s]:
                H = self.fhalo[s]
                a0 = y0 - H if self.rank > 0 else 0
                a1 = y1 + H if (self.rank < self.world - 1 and y1 < self.h) else self.h
                self.backend.blur_rows_bounded(y0, y1, self.pre[s], self.z[s], dst, a0, a1, self.missw[s][0:1])
            else:
                self.backend.blur_rows(y0, y1, self.pre[s], self.z[s], dst)
        if self.host is not None:
            self._copy_strip_to_host(s)

    # ---- pwn_tiled_submit ---------------------------------------------------------------------
    def submit(self, cam, sec=0.0):
        if self.submitted - self.delivered >= NSLOT - 1:
            raise RuntimeError("%d frames are in flight: wait() first (PWN_EBUSY)" % (NSLOT - 1))
        f = self.submitted
        s = f % NSLOT
        self.fhalo[s] = self.halo
        self.froot[s] = f % self.world if self.rotate_root else 0
        self.fcuts[s] = list(self.cuts)
        y0, y1 = self.cuts[self.rank], self.cuts[self.rank + 1]
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        plane = self.pre[s] if self.blur_passes else (self.fin[s] if (self.rank == self.froot[s] and self.host is None) else self.pre[s])
        self.missw[s].zero_()
        cost = self.backend.trace_rows(cam, float(sec), y0, y1, plane, self.z[s])
        # (in the C code the trace launch adds up its waves' lifetimes and the frame's blur moves the sum into the
        # frame's second word; a backend that measures nothing leaves it 0 and the cuts where they are)
        if cost is not None and self.blur_passes:
            self.missw[s][1] = int(cost) & 0x7fffffff
        # this frame's pre-blur rows to the neighbours, its blur, its gather: two groups, like pwn_tiled_submit
        if self.world > 1 and self.blur_passes:
            self._begin()
            if self.halo:
                H = self.halo
                if self.rank > 0:
                    self._send(self.pre[s][y0:y0 + H], self.rank - 1, TAG_HALO)
                    self._recv(self.pre[s][y0 - H:y0], self.rank - 1, TAG_HALO)
                if self.rank < self.world - 1 and y1 < self.h:
                    self._send(self.pre[s][y1 - H:y1], self.rank + 1, TAG_HALO)
                    self._recv(self.pre[s][y1:y1 + H], self.rank + 1, TAG_HALO)
            else:
                self._add_allgather(s)
            self._end()
        self._enqueue_blur(f)
        self.blurred = f + 1
        if self.world > 1:
            self._begin()
            self._add_gather(f)
            self._end()
        self.gathered = f + 1
        self.submitted = f + 1

    # ---- pwn_tiled_wait -------------------------------------------------------------------------
    def wait(self):
        """Oldest frame in flight, on every rank.  Returns (frame, redone): frame = the full frame
        tensor on the frame's root (rank 0, or rank f mod world with rotate_root; valid until five more frames were
        submitted), None elsewhere; with a host
        sink the frame in the shared host memory, on every rank."""
        if self.delivered >= self.submitted:
            raise RuntimeError("nothing in flight")
        d = self.delivered
        s = d % NSLOT
        # the ranks' words of this frame
        words = self.missv[s].clone()
        words[2 * self.rank:2 * self.rank + 2] = self.missw[s]
        words = [int(v) for v in words.cpu().tolist()]
        miss = bool(self.fhalo[s]) and any(words[2 * r] != 0 for r in range(self.world))
        cost = [words[2 * r + 1] for r in range(self.world)]
        self.last_cost = cost
        if miss:
            # every rank sees the same words and comes here together
            self.info["frames_redone"] += 1
            self.halo = 0
            self.fhalo[s] = 0
            y0, y1 = self._rows_of(s, self.rank)
            self._begin()
            self._add_allgather(s)
            self._end()
            dst = self.fin[s] if (self.rank == self.froot[s] and self.host is None) else self.out[s]
            self.backend.blur_rows(y0, y1, self.pre[s], self.z[s], dst)
            if self.host is not None:
                self._copy_strip_to_host(s)            # the strip again, and the words behind it
            self._begin()
            self._add_gather(d)
            self._end()
        # moving cuts: the same numbers on every rank, so the same new cuts, from the next submitted frame on
        if self.acc_cuts != self.fcuts[s]:
            self.acc_cuts, self.acc_cost = list(self.fcuts[s]), list(cost)
        else:
            self.acc_cost = [min(a, b) for a, b in zip(self.acc_cost, cost)]
        if self.balance_every > 0 and self.world > 1 and (d + 1) % self.balance_every == 0:
            nc = recut(self.acc_cuts, self.acc_cost, self.world, self.h, self.halo if self.halo > 0 else 8, self.max_rows)
            if nc is not None:
                self.cuts = nc
                self.recuts += 1
        self.delivered = d + 1
        self.info["frames"] += 1
        if self.host is not None:
            return self.host[s], miss                  # on every rank (valid until the second next submit)
        return (self.fin[s] if self.rank == self.froot[s] else None), miss

    def to_host(self, t):
        """uint32 numpy copy of a frame tensor."""
        return t.cpu().numpy().view(np.uint32)
