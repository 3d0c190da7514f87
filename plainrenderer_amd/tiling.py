"""Multi-GPU partition of the frame by screen rows ("bands") or into screen TILES (BASELINE config 5: 2 x 2) and the halo exchange between neighbours.

One process (or, in the single-GPU partition tests, one thread) per band. Every band holds whole-frame images and renders only its
rows (csrc/frontend/frame_pipeline.h BandSettings); where a pass reads rows of a neighbouring band the C++ host calls back into
`Exchange.run`, which moves the rows named by plrf_get_exchange_items:

  * DistTransport: torch.distributed point-to-point (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
    A band only talks to the band above and the band below; the single collective is the 512-byte histogram all-reduce.
  * LocalTransport: all bands live in one process on one GPU (threads); rows are copied device-to-device. Used to prove the
    partition bit-exact against the unpartitioned frame without a multi-GPU box.
"""
import ctypes as C
import threading

import numpy as np

BAND_ALIGNMENT = 64


def band_rows(height, n_bands, index, bounds=None):
    """Rows [begin, end) of band `index`: multiples of 64 rows (the last band ends at `height`), sizes differing by at most 64;
    or, with `bounds` (n_bands + 1 row boundaries, balanced_bounds), the rows of that partition."""
    if bounds is not None:
        return int(bounds[index]), int(bounds[index + 1])
    tiles = (height + BAND_ALIGNMENT - 1) // BAND_ALIGNMENT
    if n_bands > tiles:
        raise ValueError("more bands (%d) than 64-row tiles (%d)" % (n_bands, tiles))
    base, extra = divmod(tiles, n_bands)
    begin = index * base + min(index, extra)
    end = begin + base + (1 if index < extra else 0)
    return begin * BAND_ALIGNMENT, min(end * BAND_ALIGNMENT, height)


def equal_bounds(height, n_bands):
    return [band_rows(height, n_bands, 0)[0]] + [band_rows(height, n_bands, i)[1] for i in range(n_bands)]


def balanced_bounds(height, bounds, band_times, min_rows=2 * BAND_ALIGNMENT):
    """Static load balancing of the bands from measured per-band render times: the time of a band is spread evenly over its rows (a
    piecewise-constant cost density over the frame's rows) and the new boundaries cut that density into n equal parts, rounded to 64 rows,
    every band keeping at least `min_rows`. Deterministic: every rank computes the same partition from the same gathered times.
    -> n + 1 row boundaries."""
    n = len(band_times)
    assert len(bounds) == n + 1 and bounds[0] == 0 and bounds[-1] == height
    times = [max(float(t), 1e-9) for t in band_times]
    total = sum(times)
    cum = [0.0]
    for t in times:
        cum.append(cum[-1] + t)

    def row_at(cost):  # row where the cumulative cost reaches `cost`
        for b in range(n):
            if cost <= cum[b + 1] or b == n - 1:
                frac = (cost - cum[b]) / times[b]
                return bounds[b] + min(max(frac, 0.0), 1.0) * (bounds[b + 1] - bounds[b])
        return float(height)

    out = [0]
    for k in range(1, n):
        r = int(round(row_at(total * k / n) / BAND_ALIGNMENT)) * BAND_ALIGNMENT
        lo = out[-1] + min_rows
        hi = height - (n - k) * min_rows
        out.append(min(max(r, lo), hi))
    out.append(height)
    if any(out[i + 1] - out[i] < min(min_rows, height // n) for i in range(n)) or any(v % BAND_ALIGNMENT for v in out[1:-1]):
        return list(bounds)  # a frame too small to move boundaries in: keep the partition
    return out


class Rows:
    """what one exchange item asks for, in plain integers (mirrors plrf_exchange_item)"""

    def __init__(self, ptr, row_begin, row_end, halo_rows, row_bytes, image_rows, col_begin=0, col_end=None, image_cols=None, texel_bytes=None):
        self.ptr, self.row_begin, self.row_end, self.halo_rows, self.row_bytes, self.image_rows = ptr, row_begin, row_end, halo_rows, row_bytes, image_rows
        # tile rendering: the owned rectangle is columns [col_begin, col_end) of those rows; texels of texel_bytes bytes, image_cols per row
        self.image_cols = image_cols if image_cols is not None else 0
        self.col_begin, self.col_end = col_begin, (col_end if col_end is not None else self.image_cols)
        self.texel_bytes = texel_bytes if texel_bytes is not None else (row_bytes // image_cols if image_cols else 0)

    @classmethod
    def from_item(cls, it):
        return cls(int(it.device_ptr), int(it.row_begin), int(it.row_end), int(it.halo_rows), int(it.row_bytes), int(it.image_rows), int(it.col_begin), int(it.col_end),
                   int(it.image_cols), int(it.texel_bytes))

    # rows this band sends up / down, and the rows it receives from above / below (clipped to the image and to what exists)
    def send_up(self):
        return self.row_begin, min(self.row_begin + self.halo_rows, self.row_end)

    def send_down(self):
        return max(self.row_end - self.halo_rows, self.row_begin), self.row_end

    def recv_from_above(self):
        return max(self.row_begin - self.halo_rows, 0), self.row_begin

    def recv_from_below(self):
        return self.row_end, min(self.row_end + self.halo_rows, self.image_rows)


def neighbour_plan(rows_of_band, index, n_bands):
    """[(peer, 'send'|'recv', row0, row1)] for one item. A halo wider than the neighbouring band is clipped to that band: rows
    further away belong to the band after it and are not exchanged (stated limit: halo <= height of the neighbouring band)."""
    ops = []
    me = rows_of_band[index]
    if index > 0:
        up = rows_of_band[index - 1]
        a, b = me.send_up()
        ops.append((index - 1, "send", a, b))
        a, b = me.recv_from_above()
        a = max(a, up.row_begin)
        ops.append((index - 1, "recv", a, b))
    if index + 1 < n_bands:
        dn = rows_of_band[index + 1]
        a, b = me.send_down()
        ops.append((index + 1, "send", a, b))
        a, b = me.recv_from_below()
        b = min(b, dn.row_end)
        ops.append((index + 1, "recv", a, b))
    return [o for o in ops if o[3] > o[2]]


# ------------------------------------------------------------------ tiles: the partition as a list of rectangles (x0, y0, x1, y1), one per rank
def tile_rects(width, height, gx, gy, col_bounds=None, row_bounds=None):
    """gx x gy grid of tiles, row-major (rank = ty * gx + tx), every edge a multiple of 64 (the mirror of plrf_tile_rects)"""
    cols = list(col_bounds) if col_bounds is not None else equal_bounds(width, gx)
    rows = list(row_bounds) if row_bounds is not None else equal_bounds(height, gy)
    return [(cols[tx], rows[ty], cols[tx + 1], rows[ty + 1]) for ty in range(gy) for tx in range(gx)]


def band_rects(width, height, n_bands, bounds=None):
    """a band partition as rectangles: whole rows"""
    return [(0,) + (band_rows(height, n_bands, i, bounds)[0],) + (width,) + (band_rows(height, n_bands, i, bounds)[1],) for i in range(n_bands)]


def _scale_rect(r, frame_w, frame_h, cols, rows):
    dx, dy = max(1, (frame_w + cols // 2) // max(cols, 1)), max(1, (frame_h + rows // 2) // max(rows, 1))
    return r[0] // dx, r[1] // dy, min((r[2] + dx - 1) // dx, cols), min((r[3] + dy - 1) // dy, rows)


def _grow(r, k, w, h):
    return max(r[0] - k, 0), max(r[1] - k, 0), min(r[2] + k, w), min(r[3] + k, h)


def _intersect(a, b):
    return max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])


def rect_plan(rects, rank, frame_w, frame_h, image_cols, image_rows, halo):
    """[(peer, 'send'|'recv', x0, y0, x1, y1)] in texels of an image of image_cols x image_rows showing the frame at frame_w / image_cols scale: with every
    rank whose rectangle lies within `halo` of this rank's (the ones that touch it, edge or corner - and the one behind a neighbour narrower than the halo), the part of the own rectangle within `halo` texels of the peer's is sent, the part of the
    peer's within `halo` of the own is received (the mirror of plrf_exchange_plan_rects; for whole-row rectangles it equals neighbour_plan)"""
    ops = []
    mine = _scale_rect(rects[rank], frame_w, frame_h, image_cols, image_rows)
    for p, r in enumerate(rects):
        if p == rank:
            continue
        theirs = _scale_rect(r, frame_w, frame_h, image_cols, image_rows)
        for kind, q in (("send", _intersect(mine, _grow(theirs, halo, image_cols, image_rows))), ("recv", _intersect(theirs, _grow(mine, halo, image_cols, image_rows)))):
            if q[2] > q[0] and q[3] > q[1]:
                ops.append((p, kind) + q)
    return ops


def balanced_tile_bounds(width, height, gx, gy, col_bounds, row_bounds, tile_times, min_size=2 * BAND_ALIGNMENT):
    """static load balancing of a regular gx x gy grid from measured per-tile render times (row-major): the rows are cut by the times summed over each row
    of tiles, the columns by the times summed over each column (balanced_bounds on either axis) -> (col_bounds, row_bounds)"""
    row_times = [sum(tile_times[ty * gx + tx] for tx in range(gx)) for ty in range(gy)]
    col_times = [sum(tile_times[ty * gx + tx] for ty in range(gy)) for tx in range(gx)]
    rows = balanced_bounds(height, list(row_bounds), row_times, min_size) if gy > 1 else list(row_bounds)
    cols = balanced_bounds(width, list(col_bounds), col_times, min_size) if gx > 1 else list(col_bounds)
    return cols, rows


# ------------------------------------------------------------------ transports
class _DevMem:
    """exposes a raw device address range to torch through the CUDA array interface"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class DistTransport:
    """torch.distributed transport. device=None: the pointers are host memory (gloo CPU tests)."""

    def __init__(self, rank, world, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.rank, self.world, self.device = torch, dist, rank, world, device

    def _tensor(self, ptr, nbytes):
        if self.device is None:
            buf = (C.c_uint8 * nbytes).from_address(ptr)
            return self.torch.from_numpy(np.ctypeslib.as_array(buf))
        return self.torch.as_tensor(_DevMem(ptr, nbytes), device=self.device)

    def _on_stream(self, stream_ptr):
        if self.device is None or not stream_ptr:
            import contextlib
            return contextlib.nullcontext()
        return self.torch.cuda.stream(self.torch.cuda.ExternalStream(stream_ptr, device=self.device))

    def exchange(self, items, stream_ptr, band_meta):
        """items: [Rows] of this band; band_meta(i) -> (row_begin, row_end) scaled like items[i] for any band index."""
        self.end_exchange(self.begin_exchange(items, stream_ptr, band_meta), stream_ptr)

    def begin_exchange(self, items, stream_ptr, band_meta):
        """start the sends / receives (ordered after what is on the stream now) and return the handle end_exchange waits for"""
        dist, ops = self.dist, []
        with self._on_stream(stream_ptr):
            for i, r in enumerate(items):
                bands = [Rows(0, *band_meta(i, b), r.halo_rows, r.row_bytes, r.image_rows) for b in range(self.world)]
                bands[self.rank] = r
                for peer, kind, a, b in neighbour_plan(bands, self.rank, self.world):
                    t = self._tensor(r.ptr + a * r.row_bytes, (b - a) * r.row_bytes)
                    ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, peer))
            return dist.batch_isend_irecv(ops) if ops else []

    def end_exchange(self, handle, stream_ptr):
        with self._on_stream(stream_ptr):
            if isinstance(handle, tuple):  # a rectangle exchange: (requests, [(view, staging)]) - what arrived is scattered into the images now
                reqs, scatter = handle
                for req in reqs:
                    req.wait()
                for view, staging in scatter:
                    view.copy_(staging)
                return
            for req in handle:
                req.wait()

    def _view(self, r, x0, y0, x1, y1):
        """the rectangle of item r as a (rows, bytes) strided view of its image"""
        img = self._tensor(r.ptr, r.image_rows * r.row_bytes).view(r.image_rows, r.row_bytes)
        return img[y0:y1, x0 * r.texel_bytes:x1 * r.texel_bytes]

    def begin_exchange_rects(self, items, plans, stream_ptr):
        """tile rendering: plans[i] = rect_plan of item i. A rectangle is a strided range: it is gathered into a contiguous tensor for the send and scattered
        from one after the receive (the native exchange does the same with one pack / unpack kernel, csrc/frontend/band_exchange.cpp)"""
        dist, ops, scatter, keep = self.dist, [], [], []
        with self._on_stream(stream_ptr):
            for r, plan in zip(items, plans):
                for peer, kind, x0, y0, x1, y1 in plan:
                    view = self._view(r, x0, y0, x1, y1)
                    if kind == "send":
                        t = view.contiguous()
                        keep.append(t)
                        ops.append(dist.P2POp(dist.isend, t, peer))
                    else:
                        t = self.torch.empty_like(view, memory_format=self.torch.contiguous_format)
                        scatter.append((view, t))
                        ops.append(dist.P2POp(dist.irecv, t, peer))
            reqs = dist.batch_isend_irecv(ops) if ops else []
        self._keep = keep  # the gathered tensors live until the next exchange
        return (reqs, scatter)

    def exchange_rects(self, items, plans, stream_ptr):
        self.end_exchange(self.begin_exchange_rects(items, plans, stream_ptr), stream_ptr)

    def all_reduce_histogram(self, ptr, nbytes, stream_ptr):
        with self._on_stream(stream_ptr):
            t = self._tensor(ptr, nbytes).view(self.torch.int32)  # bin counts < 2^31: the int32 sum is the uint32 sum
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def all_reduce_depth_apex(self, ptr, nbytes, stream_ptr):
        """{min, max} of the bands' depth ranges, in place (SURVEY 8e collective 2): min on the first float, max on the second"""
        with self._on_stream(stream_ptr):
            t = self._tensor(ptr, nbytes).view(self.torch.float32)
            lo, hi = t[0:1].clone(), t[1:2].clone()
            self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
            self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
            t[0:1].copy_(lo)
            t[1:2].copy_(hi)


class LocalGroup:
    """shared state of the bands of one process (LocalTransport)"""

    def __init__(self, n_bands, lib):
        self.n, self.lib = n_bands, lib
        self.barrier = threading.Barrier(n_bands)
        self.slots = [None] * n_bands
        lib.plr_copy_device_memory.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.plr_read_device_memory.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.plr_write_device_memory.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.plr_copy_device_memory_2d.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]


class LocalTransport:
    """bands as threads of one process, each with its own (thread-local) backend on the same GPU: rows are copied device-to-device by the
    receiving band's backend. All HIP calls go through libplr (the process must not end up with a second HIP runtime)."""

    def __init__(self, group, index):
        self.g, self.rank, self.world = group, index, group.n

    def _call(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %d %s" % (what, rc, self.g.lib.plr_last_error().decode(errors="replace")))

    def _sync(self):
        self._call(self.g.lib.plr_wait_for_gpu_idle(), "plr_wait_for_gpu_idle")

    def _publish(self, payload):
        self._sync()                      # my rows are complete in HBM
        self.g.slots[self.rank] = payload
        self.g.barrier.wait()             # everybody's rows are complete and published

    def _retire(self):
        self._sync()                      # my copies are done
        self.g.barrier.wait()             # nobody still reads my rows

    def exchange(self, items, stream_ptr, band_meta):
        self._publish(items)
        for i, r in enumerate(items):
            bands = [self.g.slots[b][i] for b in range(self.world)]
            for peer, kind, a, b in neighbour_plan(bands, self.rank, self.world):
                if kind != "recv":
                    continue
                src = bands[peer].ptr + a * r.row_bytes
                self._call(self.g.lib.plr_copy_device_memory(C.c_void_p(r.ptr + a * r.row_bytes), C.c_void_p(src), C.c_size_t((b - a) * r.row_bytes)), "plr_copy_device_memory")
        self._retire()

    def begin_exchange(self, items, stream_ptr, band_meta):
        self.exchange(items, stream_ptr, band_meta)  # threads of one process: the copy is done when this returns
        return None

    def exchange_rects(self, items, plans, stream_ptr):
        """tile rendering: every received rectangle is a 2-D device copy from the owner's image (same texel grid on both sides)"""
        self._publish(items)
        for i, (r, plan) in enumerate(zip(items, plans)):
            for peer, kind, x0, y0, x1, y1 in plan:
                if kind != "recv":
                    continue
                off = y0 * r.row_bytes + x0 * r.texel_bytes
                src = self.g.slots[peer][i].ptr + off
                self._call(self.g.lib.plr_copy_device_memory_2d(C.c_void_p(r.ptr + off), C.c_size_t(r.row_bytes), C.c_void_p(src), C.c_size_t(r.row_bytes),
                                                                C.c_size_t((x1 - x0) * r.texel_bytes), C.c_size_t(y1 - y0)), "plr_copy_device_memory_2d")
        self._retire()

    def begin_exchange_rects(self, items, plans, stream_ptr):
        self.exchange_rects(items, plans, stream_ptr)
        return None

    def end_exchange(self, handle, stream_ptr):
        pass

    def all_reduce_histogram(self, ptr, nbytes, stream_ptr):
        host = np.zeros(nbytes // 4, np.uint32)
        self._call(self.g.lib.plr_read_device_memory(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nbytes)), "plr_read_device_memory")
        self.g.slots[self.rank] = host
        self.g.barrier.wait()
        total = np.sum(np.stack(self.g.slots), axis=0, dtype=np.uint64).astype(np.uint32)
        self.g.barrier.wait()
        self._call(self.g.lib.plr_write_device_memory(C.c_void_p(ptr), total.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes)), "plr_write_device_memory")


    def all_reduce_depth_apex(self, ptr, nbytes, stream_ptr):
        host = np.zeros(2, np.float32)
        self._call(self.g.lib.plr_read_device_memory(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(8)), "plr_read_device_memory")
        self.g.slots[self.rank] = host
        self.g.barrier.wait()
        both = np.stack(self.g.slots)
        total = np.array([both[:, 0].min(), both[:, 1].max()], np.float32)
        self.g.barrier.wait()
        self._call(self.g.lib.plr_write_device_memory(C.c_void_p(ptr), total.ctypes.data_as(C.c_void_p), C.c_size_t(8)), "plr_write_device_memory")


class Exchange:
    """the exchange callback of one band: glue between the C++ host's exchange points and a transport"""

    def __init__(self, fp, transport, height, n_bands, index, bounds=None, rects=None, width=None):
        """rects (+ width): the partition as rectangles, one per rank (tile rendering); else row bands (bounds / the equal partition)"""
        from .frame import EXCHANGE_HISTOGRAM
        self.fp, self.t, self.height, self.n, self.index, self.bounds = fp, transport, height, n_bands, index, bounds
        self.rects, self.width = rects, width
        self._hist_id = EXCHANGE_HISTOGRAM
        self.calls = []
        self._pending = {}
        fp.set_exchange_callback(self.run)

    def run(self, exchange_id, stream_ptr):
        from .frame import EXCHANGE_BEGIN, EXCHANGE_END, EXCHANGE_ID_MASK
        from .frame import EXCHANGE_HISTOGRAM
        if (exchange_id & EXCHANGE_ID_MASK) == EXCHANGE_HISTOGRAM:
            self.calls = []  # the histogram all-reduce is the first exchange of a frame: the log covers one frame
        self.calls.append(exchange_id)
        phase, exchange_id = exchange_id & (EXCHANGE_BEGIN | EXCHANGE_END), exchange_id & EXCHANGE_ID_MASK
        if phase == EXCHANGE_END:
            self.t.end_exchange(self._pending.pop(exchange_id), stream_ptr)
            return
        if exchange_id == self._hist_id:
            ptr, nbytes = self.fp.histogram_exchange()
            self.t.all_reduce_histogram(ptr, nbytes, stream_ptr)
            return
        from .frame import EXCHANGE_DEPTH_APEX
        if exchange_id == EXCHANGE_DEPTH_APEX:
            ptr, nbytes = self.fp.depth_apex_exchange()
            self.t.all_reduce_depth_apex(ptr, nbytes, stream_ptr)
            return
        items = [Rows.from_item(it) for it in self.fp.exchange_items(exchange_id)]
        if self.rects is not None:
            plans = [rect_plan(self.rects, self.index, self.width, self.height, r.image_cols, r.image_rows, r.halo_rows) for r in items]
            if phase == EXCHANGE_BEGIN:
                self._pending[exchange_id] = self.t.begin_exchange_rects(items, plans, stream_ptr)
            else:
                self.t.exchange_rects(items, plans, stream_ptr)
            return

        def band_meta(i, b):
            # rows of band b in item i's image: the full-resolution band scaled by the item's resolution divisor
            div = max(1, round(self.height / items[i].image_rows))
            b0, b1 = band_rows(self.height, self.n, b, self.bounds)
            return b0 // div, min((b1 + div - 1) // div, items[i].image_rows)

        # the row-band plan stops at the two neighbours (neighbour_plan): a halo taller than a neighbour would leave rows the pipeline declares valid unsent
        # (the exact mode's whole-image halo). Such an exchange goes through the rectangle plan, which reaches every band within the halo (whole-row rectangles).
        def reaches_past_a_neighbour():
            for i, r in enumerate(items):
                for b in (self.index - 1, self.index + 1):
                    if 0 <= b < self.n:
                        lo, hi = band_meta(i, b)
                        if r.halo_rows > hi - lo and ((b < self.index and lo > 0) or (b > self.index and hi < r.image_rows)):
                            return True
            return False
        if items and reaches_past_a_neighbour():
            div = max(1, round(self.height / items[0].image_rows))
            width = items[0].image_cols * div
            rects = [(0,) + (band_rows(self.height, self.n, b, self.bounds)[0],) + (width,) + (band_rows(self.height, self.n, b, self.bounds)[1],) for b in range(self.n)]
            plans = [rect_plan(rects, self.index, width, self.height, r.image_cols, r.image_rows, r.halo_rows) for r in items]
            if phase == EXCHANGE_BEGIN:
                self._pending[exchange_id] = self.t.begin_exchange_rects(items, plans, stream_ptr)
            else:
                self.t.exchange_rects(items, plans, stream_ptr)
            return
        if phase == EXCHANGE_BEGIN:
            self._pending[exchange_id] = self.t.begin_exchange(items, stream_ptr, band_meta)
        else:
            self.t.exchange(items, stream_ptr, band_meta)
