"""Row-band partition of the frame (multi-GPU path, DESIGN.md "Multi-GPU").

CPU: partition arithmetic, the neighbour exchange plan and the torch.distributed transport over gloo with world_size 2.
GPU: all bands run on ONE GPU (one host thread + one backend per band, LocalTransport) and must reproduce the unpartitioned
frame bit for bit over several frames, which exercises every dispatch base, every halo and every exchange point."""
import copy
import os
import socket
import sys
import threading

import numpy as np
import pytest

from plainrenderer_amd import tiling


# ------------------------------------------------------------------ CPU: arithmetic
def test_band_rows_partition():
    for height, n in ((4320, 4), (2160, 2), (8640, 8), (1080, 3), (192, 2), (64, 1)):
        rows = [tiling.band_rows(height, n, i) for i in range(n)]
        assert rows[0][0] == 0 and rows[-1][1] == height
        for (a0, a1), (b0, b1) in zip(rows, rows[1:]):
            assert a1 == b0
        assert all(b % 64 == 0 for b, _ in rows) and all(e % 64 == 0 for _, e in rows[:-1])
        sizes = [e - b for b, e in rows]
        assert max(sizes) - min(sizes) <= 64 + (64 - height % 64) % 64
    with pytest.raises(ValueError):
        tiling.band_rows(128, 3, 0)


def test_neighbour_plan_rows():
    mk = lambda b0, b1: tiling.Rows(0, b0, b1, 16, 8, 300)
    bands = [mk(0, 100), mk(100, 200), mk(200, 300)]
    assert tiling.neighbour_plan(bands, 0, 3) == [(1, "send", 84, 100), (1, "recv", 100, 116)]
    assert tiling.neighbour_plan(bands, 1, 3) == [(0, "send", 100, 116), (0, "recv", 84, 100), (2, "send", 184, 200), (2, "recv", 200, 216)]
    assert tiling.neighbour_plan(bands, 2, 3) == [(1, "send", 200, 216), (1, "recv", 184, 200)]
    # a halo wider than the neighbouring band is clipped to that band
    wide = [tiling.Rows(0, 0, 10, 64, 8, 40), tiling.Rows(0, 10, 20, 64, 8, 40), tiling.Rows(0, 20, 40, 64, 8, 40)]
    assert tiling.neighbour_plan(wide, 2, 3) == [(1, "send", 20, 40), (1, "recv", 10, 20)]
    # every send has the matching receive on the peer
    for bs in (bands, wide):
        for i in range(3):
            for peer, kind, a, b in tiling.neighbour_plan(bs, i, 3):
                other = "recv" if kind == "send" else "send"
                assert (i, other, a, b) in tiling.neighbour_plan(bs, peer, 3)


# ------------------------------------------------------------------ CPU: gloo world_size 2
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, height, out, bounds=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = tiling.DistTransport(rank, world, device=None)
        # two "images" (full and half resolution) holding rank-stamped rows only inside the band, garbage (0xEE) elsewhere
        items, arrays = [], []
        for div, row_bytes, halo in ((1, 24, 16), (2, 12, 8)):
            rows = height // div
            b0, b1 = tiling.band_rows(height, world, rank, bounds)
            b0, b1 = b0 // div, min((b1 + div - 1) // div, rows)
            img = np.full((rows, row_bytes), 0xEE, np.uint8)
            img[b0:b1] = ((np.arange(b0, b1)[:, None] * 3 + np.arange(row_bytes)[None, :]) % 256).astype(np.uint8) ^ 0x5A
            arrays.append(img)
            items.append(tiling.Rows(img.ctypes.data, b0, b1, halo, row_bytes, rows))

        def band_meta(i, b):
            div = (1, 2)[i]
            a0, a1 = tiling.band_rows(height, world, b, bounds)
            return a0 // div, min((a1 + div - 1) // div, height // div)

        # first image in one call, second as the overlapped form: start, "interior work" on rows the exchange does not touch, wait
        t.exchange(items[:1], None, lambda i, b: band_meta(0, b))
        handle = t.begin_exchange(items[1:], None, lambda i, b: band_meta(1, b))
        b0, b1 = items[1].row_begin, items[1].row_end
        interior = arrays[1][b0 + items[1].halo_rows:b1 - items[1].halo_rows]
        checksum = int(interior.astype(np.uint64).sum())  # reads interior rows while the transfers are in flight
        t.end_exchange(handle, None)
        assert checksum == int(interior.astype(np.uint64).sum())
        hist = np.arange(128, dtype=np.uint32) * (rank + 1)
        t.all_reduce_histogram(hist.ctypes.data, hist.nbytes, None)
        # the second collective: {min, max} of the bands' depth ranges, min on the first float, max on the second
        apex = np.array([0.25 + 0.125 * rank, 0.5 + 0.0625 * rank], np.float32)
        t.all_reduce_depth_apex(apex.ctypes.data, apex.nbytes, None)
        assert apex.tolist() == [0.25, 0.5 + 0.0625 * (world - 1)], apex
        out.put((rank, [a.copy() for a in arrays], hist.copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height,world,bounds", [(192, 2, None), (384, 3, [0, 192, 256, 384])])
def test_dist_transport_gloo_world2(height, world, bounds):
    """world 2 with the equal partition; world 3 with bands of unequal height (a balanced partition, tiling.balanced_bounds): the middle band of 64 rows
    exchanges with both neighbours"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, height, q, bounds)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, arrays, hist = q.get(timeout=120)
        got[rank] = (arrays, hist)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        arrays, hist = got[rank]
        assert np.array_equal(hist, np.arange(128, dtype=np.uint32) * (world * (world + 1) // 2))  # sum of (rank + 1)
        for (div, row_bytes, halo), img in zip(((1, 24, 16), (2, 12, 8)), arrays):
            rows = height // div
            b0, b1 = tiling.band_rows(height, world, rank, bounds)
            b0, b1 = b0 // div, min((b1 + div - 1) // div, rows)
            expect = ((np.arange(rows)[:, None] * 3 + np.arange(row_bytes)[None, :]) % 256).astype(np.uint8) ^ 0x5A
            lo, hi = max(b0 - halo, 0), min(b1 + halo, rows)
            assert np.array_equal(img[lo:hi], expect[lo:hi]), "band + halo rows hold the owners' data"
            assert (img[:lo] == 0xEE).all() and (img[hi:] == 0xEE).all(), "rows beyond the halo are untouched"


# ------------------------------------------------------------------ GPU: bands on one GPU vs the unpartitioned frame
W, H, LUT = 256, 192, 16
N_FRAMES = 3
FP_ARGS = dict(shadow_map_res=128, brdf_lut_res=LUT, froxel_depth=8, max_sdf_instances=64)


def _cams():
    from plainrenderer_amd.scene import Camera
    return [Camera.look((15.0 + 0.03 * i, -7.0, -6.0 + 0.05 * i), (0.0, 0.16, 1.0), aspect=W / H) for i in range(N_FRAMES + 1)]


def _make_inputs():
    from plainrenderer_amd import synth
    from plainrenderer_amd.frame import SyntheticInputs
    cams = _cams()
    scene = synth.SynthScene(grid=4, cell=8.0, seed_id=800)
    return SyntheticInputs(scene, cams[1], cams[0], W, H, sdf_res=16, shadow_res=128, froxel_depth=8, sun_direction=(0.35, -0.8, 0.45))


def _run(inputs, exact, band=None, group=None, halos=None, out=None, half_res=1, extra=None, rects=None, transport="python"):
    """one backend + pipeline on the calling thread; band = (index, n) or None for the whole frame; extra = more FramePipeline settings;
    rects: the partition as rectangles (tile rendering) - band[0] then indexes it"""
    from plainrenderer_amd import RenderBackend
    from plainrenderer_amd.frame import FramePipeline
    try:
        be = RenderBackend(W, H, device=0)
        be.setMathMode(not exact)
        kw = dict(FP_ARGS, sdf_half_res_trace=half_res, **(extra or {}))
        if band is not None and rects is not None:
            x0, y0, x1, y1 = rects[band[0]]
            kw.update(band_row_begin=y0, band_row_end=y1, band_col_begin=x0, band_col_end=x1, **(halos or {}))
        elif band is not None:
            b0, b1 = tiling.band_rows(H, band[1], band[0])
            kw.update(band_row_begin=b0, band_row_end=b1, **(halos or {}))
        fp = FramePipeline(be, W, H, **kw)
        inp = copy.copy(inputs)
        inp.upload(fp)
        ex = None
        if band is not None and transport == "native":
            # the native exchange (csrc/frontend/band_exchange.cpp) over its in-process transport: the exchange code of the multi-GPU run, device copies for links
            fp.attach_local_rects(group, band[0], band[1], W, H, rects if rects is not None else tiling.band_rects(W, H, band[1]))
        elif band is not None:
            ex = tiling.Exchange(fp, tiling.LocalTransport(group, band[0]), H, band[1], band[0], rects=rects, width=W)
        cams = _cams()
        frames = []
        for f in range(N_FRAMES):
            fp.frame(cams[f + 1], 1.0 / 60.0, 0.5 + f / 60.0)
            cur = (f + 1) % 2
            frames.append(dict(post=be.downloadImage(fp.image("post1"), 0, np.uint32).reshape(H, W).copy(),
                               color=be.downloadImage(fp.image("color%d" % cur), 0, np.uint32).reshape(H, W).copy(),
                               swap=be.downloadImage(fp.image("swapchain"), 0, np.uint8).reshape(H, W, 4).copy(),
                               hist=be.downloadStorageBuffer(fp.storage_buffer("histogram"), 512, dtype=np.uint32).copy(),
                               light=be.downloadStorageBuffer(fp.storage_buffer("light"), 20, dtype=np.uint8).tobytes(),
                               shadow=be.downloadStorageBuffer(fp.storage_buffer("sunShadowInfo"), 304, dtype=np.uint8).tobytes()))
        res = dict(frames=frames, calls=list(ex.calls) if ex else [], fused=be.getPassFusion()[1], edge_signal=be.getEdgeSignal(),
                   exchange=(fp.rccl_stats(), fp.rccl_info()) if (band is not None and transport == "native") else None)
        fp.destroy()
        be.shutdown()
        out[band[0] if band is not None else "full"] = res
    except BaseException as e:  # surface failures of worker threads (and unblock the others)
        out[band[0] if band is not None else "full"] = e
        if group is not None:
            group.abort() if transport == "native" else group.barrier.abort()
        raise


def _run_full(inputs, exact, half_res=1, extra=None):
    # on its own thread = its own backend: the session-wide test backend may already live on the main thread
    out = {}
    t = threading.Thread(target=_run, args=(inputs, exact), kwargs=dict(out=out, half_res=half_res, extra=extra))
    t.start()
    t.join(timeout=600)
    assert "full" in out, "the unpartitioned frame did not finish"
    if isinstance(out["full"], BaseException):
        raise out["full"]
    return out["full"]


def _run_bands(inputs, n, exact, halos, half_res=1, extra=None, rects=None, transport="python"):
    from plainrenderer_amd import backend
    from plainrenderer_amd.frame import LocalExchangeGroup
    group = LocalExchangeGroup(n) if transport == "native" else tiling.LocalGroup(n, backend._load())
    out = {}
    threads = [threading.Thread(target=_run, args=(inputs, exact, (i, n), group, halos, out, half_res, extra, rects, transport)) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    if transport == "native":
        group.destroy()
    for i in range(n):
        assert i in out, "band %d did not finish" % i
        if isinstance(out[i], BaseException):
            raise out[i]
    return out


def _compare(full, bands, n, what=("post", "color", "swap"), rects=None):
    mism = {}
    for f in range(N_FRAMES):
        for i in range(n):
            if rects is not None:
                x0, b0, x1, b1 = rects[i]
            else:
                x0, (b0, b1), x1 = 0, tiling.band_rows(H, n, i), W
            fr, bf = full["frames"][f], bands[i]["frames"][f]
            assert np.array_equal(fr["hist"], bf["hist"]) and int(bf["hist"].sum()) == W * H, "histogram frame %d band %d" % (f, i)
            assert fr["light"] == bf["light"], "exposure frame %d band %d" % (f, i)
            for k in what:
                d = fr[k][b0:b1, x0:x1] != bf[k][b0:b1, x0:x1]
                mism[(f, i, k)] = float(d.mean())
    return mism


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["native", "python"])
@pytest.mark.parametrize("half_res", [1, 0])
def test_gpu_two_bands_reproduce_the_full_frame_bit_exact(half_res, transport):
    # n = 2: the neighbour is the rest of the frame, so halos as tall as the image make every pass's inputs complete and the
    # partitioned frame must equal the unpartitioned one in every bit, over 3 frames of temporal feedback
    inputs = _make_inputs()
    full = _run_full(inputs, True, half_res)
    # half_res 1: the default overlapped exchange (producers' edge rows first, begin / end callbacks); half_res 0: one callback per exchange
    overlap = half_res == 1
    halos = dict(band_gi_halo=H, band_gi_history_halo=H, band_post_halo=H, band_taa_history_halo=H, band_overlap_exchange=int(overlap))
    bands = _run_bands(inputs, 2, True, halos, half_res, transport=transport)
    mism = _compare(full, bands, 2)
    bad = {k: v for k, v in mism.items() if v != 0.0}
    assert not bad, bad
    B, E = 0x100, 0x200
    expected = [0, 1 | B, 1 | E, 2 | B, 2 | E, 3, 4 | B, 4 | E] if overlap else [0, 1, 2, 3, 4]
    if transport == "python":  # (the native exchange is the callback itself: no Python-side log of its calls)
        assert bands[0]["calls"][:len(expected)] == expected


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [(2, 2), (4, 3)])
def test_gpu_light_matrices_of_tiles_equal_the_unpartitioned_ones(grid):
    """the same for screen tiles: a tile reduces ITS columns of the per-tile pyramid's top level. 4 x 3 tiles of 64 x 64 pixels are one texel of that level each - the
    tile at column 0 used to read "one workgroup from column 0" as whole rows and reduced texels it never built; the column range is explicit now"""
    inputs = _make_inputs()
    extra = dict(run_light_matrix=1)
    full = _run_full(inputs, True, extra=extra)
    rects = _tile_rects(*grid)
    n = len(rects)
    big = max(W, H)
    halos = dict(band_gi_halo=big, band_gi_history_halo=big, band_post_halo=big, band_taa_history_halo=big)
    tiles = _run_bands(inputs, n, True, halos, extra=extra, rects=rects, transport="native")
    for i in range(n):
        for f in range(N_FRAMES):
            assert tiles[i]["frames"][f]["shadow"] == full["frames"][f]["shadow"], "cascade fit of tile %d differs in frame %d" % (i, f)
    assert full["frames"][0]["shadow"] != bytes(304)
    mism = _compare(full, tiles, n, rects=rects)
    bad = {k: v for k, v in mism.items() if v != 0.0}
    assert not bad, bad


@pytest.mark.gpu
def test_gpu_light_matrices_of_bands_equal_the_unpartitioned_ones():
    # SURVEY 8e, collective 2: lightMatrix.comp fits the shadow cascades to the depth range of the FRAME (the apex of the depth
    # pyramid). A band reduces its rows of the per-tile pyramid and the bands' ranges are all-reduced (min, max): every band must end up with the
    # cascade splits and light matrices of the unpartitioned frame, bit for bit, and so must the frames shaded with them
    inputs = _make_inputs()
    extra = dict(run_light_matrix=1)
    full = _run_full(inputs, True, extra=extra)
    halos = dict(band_gi_halo=H, band_gi_history_halo=H, band_post_halo=H, band_taa_history_halo=H)
    bands = _run_bands(inputs, 3, True, halos, extra=extra)
    from plainrenderer_amd.frame import EXCHANGE_DEPTH_APEX
    for i in range(3):
        assert EXCHANGE_DEPTH_APEX in bands[i]["calls"], "band %d never all-reduced its depth range" % i
        for f in range(N_FRAMES):
            assert bands[i]["frames"][f]["shadow"] == full["frames"][f]["shadow"], "cascade fit of band %d differs in frame %d" % (i, f)
    assert full["frames"][0]["shadow"] != bytes(304)
    mism = _compare(full, bands, 3)
    # (three bands of 64 rows with whole-image halos: the GI of a band two bands away is not exchanged, so only the direct light is held to bits here)
    assert max(v for (f, i, k), v in mism.items() if k == "color") < 0.05, mism


@pytest.mark.gpu
def test_gpu_three_bands_default_halos():
    # default halos (64 trace rows GI, 16 history, 224 post): bands of 64 rows here - smaller than the halos, which are clipped to the neighbouring
    # band; pixels whose 1.5 m denoiser disc or bloom footprint reaches past the neighbouring band may differ - they must be few
    inputs = _make_inputs()
    full = _run_full(inputs, True)
    bands = _run_bands(inputs, 3, True, None)
    mism = _compare(full, bands, 3)
    assert max(v for (f, i, k), v in mism.items() if k == "color") < 0.05, mism
    assert max(mism.values()) < 0.6, mism
    # splitting the producers into edge and interior dispatches (overlapped exchange, the default above) changes nothing
    plain = _run_bands(inputs, 3, True, dict(band_overlap_exchange=0))
    for i in range(3):
        for f in range(N_FRAMES):
            for k in ("post", "color", "swap"):
                assert np.array_equal(bands[i]["frames"][f][k], plain[i]["frames"][f][k]), "overlapped vs plain exchange: band %d frame %d %s" % (i, f, k)


@pytest.mark.gpu
def test_gpu_overlapped_exchange_with_interior_rows_equals_plain_exchange():
    # halos much smaller than the bands: every producer of an exchanged image is recorded as top edge, bottom edge, exchange start,
    # interior; the frames must equal those of the one-callback-per-exchange recording in every bit (fast kernels, 3 frames)
    inputs = _make_inputs()
    small = dict(band_gi_halo=8, band_gi_history_halo=8, band_post_halo=16, band_taa_history_halo=8)
    over = _run_bands(inputs, 2, False, dict(small, band_overlap_exchange=1))
    plain = _run_bands(inputs, 2, False, dict(small, band_overlap_exchange=0))
    B, E = 0x100, 0x200
    assert over[0]["calls"][:8] == [0, 1 | B, 1 | E, 2 | B, 2 | E, 3, 4 | B, 4 | E] and plain[0]["calls"][:5] == [0, 1, 2, 3, 4]
    for i in range(2):
        b0, b1 = tiling.band_rows(H, 2, i)
        for f in range(N_FRAMES):
            for k in ("post", "color", "swap"):
                assert np.array_equal(over[i]["frames"][f][k][b0:b1], plain[i]["frames"][f][k][b0:b1]), "band %d frame %d %s" % (i, f, k)


@pytest.mark.gpu
def test_gpu_middle_band_launches_both_edges_at_once():
    """three bands, halos smaller than the bands: the middle band records every producer of an exchanged image as top edge, bottom edge, exchange start,
    interior; the backend issues the two edge dispatches of the trace, the temporal GI filter and the TAA resolve as ONE launch each over two row ranges
    (pass fusion, backend.h launchOverTwoRowRanges). Frames are the same bits as with the plain exchange (one dispatch per producer)."""
    inputs = _make_inputs()
    small = dict(band_gi_halo=8, band_gi_history_halo=8, band_post_halo=16, band_taa_history_halo=8)
    over = _run_bands(inputs, 3, False, dict(small, band_overlap_exchange=1))
    plain = _run_bands(inputs, 3, False, dict(small, band_overlap_exchange=0))
    # the outer bands have one edge each; the middle band fuses 3 x 2 edge executions on top of whatever both recordings fuse
    assert over[1]["fused"] >= plain[1]["fused"] + 6, (over[1]["fused"], plain[1]["fused"])
    for i in range(3):
        b0, b1 = tiling.band_rows(H, 3, i)
        for f in range(N_FRAMES):
            for k in ("post", "color", "swap"):
                assert np.array_equal(over[i]["frames"][f][k][b0:b1], plain[i]["frames"][f][k][b0:b1]), "band %d frame %d %s" % (i, f, k)


@pytest.mark.gpu
def test_gpu_rows_first_producers_raise_the_edge_signal_and_change_no_bit():
    """band_overlap_exchange 2 (the default): a producer of an exchanged image is ONE launch that writes the rows its neighbours need first and raises
    the backend's edge signal when they are complete (plr.h first_rows) - no edge / interior split. Same bits as the plain exchange; the signal word
    holds the value of the last rows-first launch (three per frame and band: trace, temporal GI filter, TAA resolve)."""
    inputs = _make_inputs()
    small = dict(band_gi_halo=8, band_gi_history_halo=8, band_post_halo=16, band_taa_history_halo=8)
    first = _run_bands(inputs, 3, False, dict(small, band_overlap_exchange=2))
    plain = _run_bands(inputs, 3, False, dict(small, band_overlap_exchange=0))
    B, E = 0x100, 0x200
    assert first[1]["calls"][:8] == [0, 1 | B, 1 | E, 2 | B, 2 | E, 3, 4 | B, 4 | E]
    for i in range(3):
        ptr, value, now = first[i]["edge_signal"]
        assert ptr is not None, "no stream memory operations on this platform?"
        assert value == 3 * N_FRAMES and now == value, (i, value, now)  # every rows-first launch raised the signal to its value, in order
        assert plain[i]["edge_signal"][1] == 0
        b0, b1 = tiling.band_rows(H, 3, i)
        for f in range(N_FRAMES):
            for k in ("post", "color", "swap"):
                assert np.array_equal(first[i]["frames"][f][k][b0:b1], plain[i]["frames"][f][k][b0:b1]), "band %d frame %d %s" % (i, f, k)


@pytest.mark.gpu
def test_gpu_two_bands_fast_math_matches_full_frame():
    # the default (PLR_MATH_FAST) kernel set through the same partition: fast kernels are deterministic too
    inputs = _make_inputs()
    full = _run_full(inputs, False)
    halos = dict(band_gi_halo=H, band_gi_history_halo=H, band_post_halo=H, band_taa_history_halo=H)
    bands = _run_bands(inputs, 2, False, halos)
    mism = _compare(full, bands, 2)
    bad = {k: v for k, v in mism.items() if v != 0.0}
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["rccl-c++", "torch-python"])
def test_gpu_bench_band_path_over_rccl_group_of_one(transport):
    """bench.py's multi-GPU code path with a group of size 1: the band pipeline, the C++ host's RCCL exchange (ncclCommInitRank, the histogram
    ncclAllReduce, the send / receive groups - empty for a single band) or, with --python-exchange, the torch.distributed transport"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--force-bands", "--steps", "3", "--warmup", "1", "--profile-frames", "2", "--no-cpu-baseline",
           "--width", "512", "--height", "256", "--grid", "4", "--sdf-res", "16", "--shadow-res", "256"] + (["--python-exchange"] if transport == "torch-python" else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["value"] > 0 and any(k.startswith("Exchange:") for k in out["passes_ms"]), out["passes_ms"]
    if transport == "rccl-c++":
        assert out["exchange"]["point_to_point_groups_per_frame"] >= 3 and out["exchange"]["rank0_bytes_sent_per_frame"] == 0


@pytest.mark.gpu
def test_gpu_bench_refuses_a_world_size_it_was_not_asked_for():
    """`--gpus 2` under WORLD_SIZE=1 (or on a one-GPU box) must fail loudly, not measure a single GPU and label it 2"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_gpu_rccl_transport_moves_rows_through_the_overlapped_exchange_sequence(backend):
    """One GPU cannot host two RCCL ranks, so the C++ transport's mechanics (group of ncclSend / ncclRecv on the communication stream, ordered
    against the launch stream by events) are exercised with this rank as its own peer: rows [4, 12) of an image arrive on rows [40, 48)."""
    from plainrenderer_amd.frame import FramePipeline
    w, h = 256, 128
    fp = FramePipeline(backend, w, h, shadow_map_res=128, brdf_lut_res=16, froxel_depth=8, max_sdf_instances=16, band_row_begin=0, band_row_end=h)
    try:
        fp.attach_rccl(fp.rccl_unique_id(), 0, 1, h)
        img = fp.image("post1")
        data = np.random.default_rng(3).integers(0, 2 ** 32, (h, w), dtype=np.uint32)
        backend.uploadImage(img, data)
        ptr, size = backend.imageDevicePointer(img, 0)
        assert size == w * h * 4
        fp.rccl_self_test(ptr, w * 4, 4, 40, 8)
        got = backend.downloadImage(img, 0, np.uint32).reshape(h, w)
        expect = data.copy()
        expect[40:48] = data[4:12]
        assert np.array_equal(got, expect)
    finally:
        fp.destroy()


def test_cpp_exchange_plan_equals_the_python_plan():
    """the C++ RCCL exchange (csrc/frontend/band_exchange.cpp) plans its ncclSend / ncclRecv with plrf_band_rows / plrf_exchange_plan; the gloo
    test above exercises the Python plan - both must describe the same transfers for every band count, image scale and halo"""
    import ctypes as C
    from plainrenderer_amd import backend, tiling

    class Op(C.Structure):
        _fields_ = [("peer", C.c_uint32), ("send", C.c_uint32), ("row_begin", C.c_uint32), ("row_end", C.c_uint32)]
    lib = C.CDLL(backend.LIB_PATH)
    for height in (4320, 2160, 1088, 8640, 200):
        for n in (1, 2, 3, 4, 8):
            if n > (height + 63) // 64:
                continue
            for b in range(n):
                r0, r1 = C.c_uint32(), C.c_uint32()
                assert lib.plrf_band_rows(C.c_uint32(height), C.c_uint32(n), C.c_uint32(b), C.byref(r0), C.byref(r1)) == 0
                assert (r0.value, r1.value) == tiling.band_rows(height, n, b)
            for div in (1, 2):
                image_rows = height // div
                for halo in (0, 8, 16, 32, 64, 320, 5000):
                    def rows_of(band):
                        b0, b1 = tiling.band_rows(height, n, band)
                        return tiling.Rows(0, b0 // div, min((b1 + div - 1) // div, image_rows), halo, 16, image_rows)
                    bands = [rows_of(k) for k in range(n)]
                    for b in range(n):
                        ops = (Op * 4)()
                        cnt = C.c_uint32()
                        me = bands[b]
                        assert lib.plrf_exchange_plan(C.c_uint32(height), C.c_uint32(n), C.c_uint32(b), C.c_uint32(image_rows), C.c_uint32(halo), C.c_uint32(me.row_begin),
                                                      C.c_uint32(me.row_end), ops, C.byref(cnt)) == 0
                        got = [(int(o.peer), "send" if o.send else "recv", int(o.row_begin), int(o.row_end)) for o in ops[:cnt.value]]
                        assert got == tiling.neighbour_plan(bands, b, n), (height, n, b, div, halo)
    # what one rank sends, its neighbour receives: the plans of two adjacent bands mirror each other
    height, n, halo = 4320, 4, 218
    bands = [tiling.Rows(0, *tiling.band_rows(height, n, k), halo, 16, height) for k in range(n)]
    for b in range(n - 1):
        down = [o for o in tiling.neighbour_plan(bands, b, n) if o[0] == b + 1]
        up = [o for o in tiling.neighbour_plan(bands, b + 1, n) if o[0] == b]
        assert {(k, a, e) for _, k, a, e in down} == {("recv" if k == "send" else "send", a, e) for _, k, a, e in up}


def test_balanced_partition_and_its_exchange_plan():
    """tiling.balanced_bounds (bench.py's static load balancing): boundaries stay multiples of 64 and strictly increasing, a band that took longer
    gets fewer rows, equal times are a fixed point; the C++ plan (plrf_exchange_plan_rows) of such a partition equals the Python plan"""
    import ctypes as C
    from plainrenderer_amd import backend, tiling

    class Op(C.Structure):
        _fields_ = [("peer", C.c_uint32), ("send", C.c_uint32), ("row_begin", C.c_uint32), ("row_end", C.c_uint32)]
    lib = C.CDLL(backend.LIB_PATH)
    height, n = 4320, 4
    eq = tiling.equal_bounds(height, n)
    assert eq == [0, 1088, 2176, 3264, 4320]
    bal = tiling.balanced_bounds(height, eq, [0.788, 1.073, 1.172, 1.113])  # the measured band times of the 8K frame (profiles/r02n_band_cost.txt)
    assert bal[0] == 0 and bal[-1] == height and all(b % 64 == 0 for b in bal[1:-1]) and all(bal[i + 1] > bal[i] for i in range(n))
    assert bal[1] - bal[0] > eq[1] - eq[0] and bal[3] - bal[2] < eq[3] - eq[2]  # the cheap sky band grows, the slowest band shrinks
    assert tiling.balanced_bounds(height, bal, [1.0] * n) == bal
    assert tiling.balanced_bounds(128, [0, 64, 128], [1.0, 5.0]) == [0, 64, 128]  # nothing to move in a two-tile frame
    assert tiling.band_rows(height, n, 2, bal) == (bal[2], bal[3])
    for div in (1, 2):
        image_rows = height // div
        for halo in (16, 64, 224, 2000):
            bands = [tiling.Rows(0, bal[k] // div, min((bal[k + 1] + div - 1) // div, image_rows), halo, 16, image_rows) for k in range(n)]
            for b in range(n):
                ops, cnt = (Op * 4)(), C.c_uint32()
                rows = (C.c_uint32 * (n + 1))(*bal)
                assert lib.plrf_exchange_plan_rows(C.c_uint32(height), C.c_uint32(n), rows, C.c_uint32(b), C.c_uint32(image_rows), C.c_uint32(halo),
                                                   C.c_uint32(bands[b].row_begin), C.c_uint32(bands[b].row_end), ops, C.byref(cnt)) == 0
                got = [(int(o.peer), "send" if o.send else "recv", int(o.row_begin), int(o.row_end)) for o in ops[:cnt.value]]
                assert got == tiling.neighbour_plan(bands, b, n), (b, div, halo)
    bad = (C.c_uint32 * (n + 1))(0, 1000, 2176, 3264, 4320)  # an interior boundary that is not a multiple of 64 is refused
    assert lib.plrf_exchange_plan_rows(C.c_uint32(height), C.c_uint32(n), bad, C.c_uint32(0), C.c_uint32(height), C.c_uint32(16), C.c_uint32(0), C.c_uint32(1000), (Op * 4)(),
                                       C.byref(C.c_uint32())) != 0


@pytest.mark.gpu
def test_gpu_two_bands_of_realistic_height_with_default_halos(monkeypatch):
    """Two bands of 576 rows of a 1024 x 1152 frame with the DEFAULT halos (64 trace rows of GI, 16 of GI history, 224 rows of resolved colour,
    32 of TAA history), benchmarked kernel set, three frames of temporal feedback, against the unpartitioned frame. Everything except the
    spatial GI filter is exact for these halos; the filter's world-space disc reaches past 64 trace rows on near geometry, where a band gives the
    sample weight 0 (the shader's off-screen rule) instead of the neighbour's texel - the stated deviation of band rendering. Its size:"""
    import sys
    import parity
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "W", 1024)
    monkeypatch.setattr(mod, "H", 1152)
    inputs = _make_inputs()  # (shadow cascades 128^2, 8 froxel slices: FP_ARGS)
    full = _run_full(inputs, False)
    bands = _run_bands(inputs, 2, False, None)
    worst_within1, worst_mean = 1.0, 0.0
    for f in range(N_FRAMES):
        for i in range(2):
            b0, b1 = tiling.band_rows(1152, 2, i)
            fr, bf = full["frames"][f], bands[i]["frames"][f]
            assert np.array_equal(fr["hist"], bf["hist"]), "histogram all-reduce: identical exposure input on every band (frame %d)" % f
            d = parity.r11g11b10_code_diff(bf["post"][b0:b1], fr["post"][b0:b1])
            within1 = float((d <= 1).all(axis=1).mean())
            a, b = pixfmt_unpack(bf["post"][b0:b1]), pixfmt_unpack(fr["post"][b0:b1])
            mean_rel = float(np.abs(a - b).mean() / b.mean())
            print("BANDS frame %d band %d: pixels within one code of the unpartitioned frame %.5f, max code diff %d, mean rel err %.2e" % (f, i, within1, int(d.max()), mean_rel))
            worst_within1, worst_mean = min(worst_within1, within1), max(worst_mean, mean_rel)
    assert worst_within1 >= 0.995 and worst_mean <= 1e-3  # measured on MI355X: 1.00000 of the pixels within one code in every band and frame


def pixfmt_unpack(a):
    from plainrenderer_amd import pixfmt
    return pixfmt.unpack_r11g11b10(np.ascontiguousarray(a).reshape(-1))


# ------------------------------------------------------------------ tile rendering (round 5): the frame as a grid of screen tiles, BASELINE config 5's 2 x 2
def test_rect_plan_of_bands_equals_the_band_plan_and_the_cpp_plan():
    """tiling.rect_plan / plrf_exchange_plan_rects describe a partition into rectangles. For whole-row rectangles they must be the band plan
    (neighbour_plan / plrf_exchange_plan_rows) transfer for transfer; for tile grids the C++ plan equals the Python one, every send has its receive on the
    peer, and what a tile receives is exactly its halo frame inside the image (corners included)"""
    import ctypes as C
    from plainrenderer_amd import backend

    class RectOp(C.Structure):
        _fields_ = [(n, C.c_uint32) for n in ("peer", "send", "x0", "y0", "x1", "y1")]
    lib = C.CDLL(backend.LIB_PATH)

    def cpp_plan(w, h, rects, rank, cols, rows, halo):
        n = len(rects)
        ops, cnt = (RectOp * (2 * n))(), C.c_uint32()
        flat = (C.c_uint32 * (4 * n))(*[v for r in rects for v in r])
        assert lib.plrf_exchange_plan_rects(C.c_uint32(w), C.c_uint32(h), C.c_uint32(n), flat, C.c_uint32(rank), C.c_uint32(cols), C.c_uint32(rows), C.c_uint32(halo), ops, C.c_uint32(2 * n),
                                            C.byref(cnt)) == 0
        return [(int(o.peer), "send" if o.send else "recv", int(o.x0), int(o.y0), int(o.x1), int(o.y1)) for o in ops[:cnt.value]]

    class RowOp(C.Structure):
        _fields_ = [(n, C.c_uint32) for n in ("peer", "send", "row_begin", "row_end")]

    def cpp_band_plan(h, rects, rank, rows, halo, own):
        # what the native exchange moves for a band (planBandRows): every band within the halo's reach
        n = len(rects)
        ops, cnt = (RowOp * (2 * n))(), C.c_uint32()
        bounds = (C.c_uint32 * (n + 1))(*([r[1] for r in rects] + [h]))
        assert lib.plrf_exchange_plan_all_bands(C.c_uint32(h), C.c_uint32(n), bounds, C.c_uint32(rank), C.c_uint32(rows), C.c_uint32(halo), C.c_uint32(own[0]), C.c_uint32(own[1]), ops,
                                                C.c_uint32(2 * n), C.byref(cnt)) == 0, lib.plrf_rccl_last_error()
        return [(int(o.peer), "send" if o.send else "recv", int(o.row_begin), int(o.row_end)) for o in ops[:cnt.value]]

    # bands as rectangles
    w = 7680
    for height, n in ((4320, 4), (2160, 2), (1088, 3), (8640, 8)):
        rects = tiling.band_rects(w, height, n)
        for div in (1, 2):
            cols, rows = w // div, height // div
            for halo in (0, 16, 64, 224, 5000):
                bands = [tiling.Rows(0, r[1] // div, min((r[3] + div - 1) // div, rows), halo, 16 * cols, rows) for r in rects]
                for b in range(n):
                    got = tiling.rect_plan(rects, b, w, height, cols, rows, halo)
                    assert cpp_plan(w, height, rects, b, cols, rows, halo) == got
                    assert [(p, k, 0, a, cols, e) for p, k, a, e in cpp_band_plan(height, rects, b, rows, halo, (bands[b].row_begin, bands[b].row_end))] == got
                    if halo <= min(e.row_end - e.row_begin for e in bands):
                        assert got == [(peer, kind, 0, a, cols, e) for peer, kind, a, e in tiling.neighbour_plan(bands, b, n)], (height, n, b, div, halo)
                    else:  # a halo wider than a neighbour reaches the band behind it (the band plan stops at the neighbour)
                        if halo >= rows:
                            assert sorted((p, k) for p, k, *_ in got) == sorted((p, k) for p in range(n) if p != b for k in ("send", "recv"))
                        lo, hi = max(bands[b].row_begin - halo, 0), min(bands[b].row_end + halo, rows)
                        assert sum((y1 - y0) for p, k, x0, y0, x1, y1 in got if k == "recv") == (hi - lo) - (bands[b].row_end - bands[b].row_begin)
    # tile grids
    for (fw, fh, gx, gy) in ((7680, 4320, 2, 2), (7680, 8640, 2, 4), (3840, 2160, 4, 2), (256, 192, 2, 2), (7680, 2160, 2, 1)):
        rects = tiling.tile_rects(fw, fh, gx, gy)
        flat = (C.c_uint32 * (4 * gx * gy))()
        assert lib.plrf_tile_rects(C.c_uint32(fw), C.c_uint32(fh), C.c_uint32(gx), C.c_uint32(gy), None, None, flat) == 0
        assert [tuple(flat[4 * i:4 * i + 4]) for i in range(gx * gy)] == rects
        assert sum((r[2] - r[0]) * (r[3] - r[1]) for r in rects) == fw * fh
        for div in (1, 2):
            cols, rows = fw // div, fh // div
            # PLRF_HALO_WHOLE_IMAGE passed straight to the public plan function saturates (it used to wrap in 32 bits: ADVICE r05): the plan of a halo as large as the image
            for k in range(gx * gy):
                assert cpp_plan(fw, fh, rects, k, cols, rows, 0xffffffff) == cpp_plan(fw, fh, rects, k, cols, rows, max(cols, rows)) == tiling.rect_plan(rects, k, fw, fh, cols, rows, max(cols, rows))
            for halo in (8, 16, 128, 224):
                plans = [tiling.rect_plan(rects, k, fw, fh, cols, rows, halo) for k in range(gx * gy)]
                for k, plan in enumerate(plans):
                    assert cpp_plan(fw, fh, rects, k, cols, rows, halo) == plan, (fw, fh, gx, gy, k, div, halo)
                    got = np.zeros((rows, cols), np.int32)
                    for peer, kind, x0, y0, x1, y1 in plan:
                        other = "recv" if kind == "send" else "send"
                        assert (k, other, x0, y0, x1, y1) in plans[peer]
                        if kind == "recv":
                            got[y0:y1, x0:x1] += 1
                    mx0, my0, mx1, my1 = tiling._scale_rect(rects[k], fw, fh, cols, rows)
                    want = np.zeros((rows, cols), np.int32)
                    want[max(my0 - halo, 0):my1 + halo, max(mx0 - halo, 0):mx1 + halo] = 1
                    want[my0:my1, mx0:mx1] = 0
                    if halo <= min(r[2] - r[0] for r in rects) // div and halo <= min(r[3] - r[1] for r in rects) // div:  # (a halo wider than a neighbour is clipped to it)
                        assert np.array_equal(got, want), "a tile receives its halo frame, every texel once (%s)" % ((fw, fh, gx, gy, k, div, halo),)
    # rectangles that overlap / leave a hole / sit off the 64-pixel grid are refused
    bad = (C.c_uint32 * 8)(0, 0, 100, 192, 100, 0, 256, 192)
    assert lib.plrf_exchange_plan_rects(C.c_uint32(256), C.c_uint32(192), C.c_uint32(2), bad, C.c_uint32(0), C.c_uint32(256), C.c_uint32(192), C.c_uint32(8), (RectOp * 4)(), C.c_uint32(4),
                                        C.byref(C.c_uint32())) != 0


def _gloo_tile_worker(rank, world, port, fw, fh, gx, gy, out, images=((1, 4, 16), (2, 8, 8))):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = tiling.DistTransport(rank, world, device=None)
        rects = tiling.tile_rects(fw, fh, gx, gy)
        items, arrays, plans = [], [], []
        for div, texel_bytes, halo in images:
            cols, rows = fw // div, fh // div
            x0, y0, x1, y1 = tiling._scale_rect(rects[rank], fw, fh, cols, rows)
            img = np.full((rows, cols * texel_bytes), 0xEE, np.uint8)
            stamp = ((np.arange(rows)[:, None] * 7 + np.arange(cols * texel_bytes)[None, :] * 3) % 251).astype(np.uint8)
            img[y0:y1, x0 * texel_bytes:x1 * texel_bytes] = stamp[y0:y1, x0 * texel_bytes:x1 * texel_bytes]
            arrays.append(img)
            items.append(tiling.Rows(img.ctypes.data, y0, y1, halo, cols * texel_bytes, rows, x0, x1, cols, texel_bytes))
            plans.append(tiling.rect_plan(rects, rank, fw, fh, cols, rows, halo))
        t.exchange_rects(items[:1], plans[:1], None)                                    # one call
        t.end_exchange(t.begin_exchange_rects(items[1:], plans[1:], None), None)        # the overlapped form
        out.put((rank, [a.copy() for a in arrays]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gx,gy,images", [(2, 2, ((1, 4, 16), (2, 8, 8))), (1, 4, ((1, 4, 4096), (2, 8, 4096))), (2, 2, ((1, 4, 4096), (2, 8, 40)))],
                         ids=["tiles2x2", "bands4-whole-image-halo", "tiles2x2-whole-image-halo"])
def test_dist_transport_gloo_world4_tiles(gx, gy, images):
    """the torch.distributed transport over a partition into four rectangles, world size 4 over gloo: every rank ends up with its own texels, the halo frame of the
    other ranks' texels (corners from the diagonal tile) and nothing else. With a halo as large as the image (the exact mode of a partitioned frame) every rank is
    a peer - a band receives the rows of the band BEHIND its neighbour too - and every rank ends up with the whole image."""
    import torch.multiprocessing as mp
    fw, fh, world = 256, 256 if gy == 4 else 128, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_tile_worker, args=(r, world, port, fw, fh, gx, gy, q, images)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, arrays = q.get(timeout=180)
        got[rank] = arrays
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rects = tiling.tile_rects(fw, fh, gx, gy)
    for rank in range(world):
        for (div, texel_bytes, halo), img in zip(images, got[rank]):
            cols, rows = fw // div, fh // div
            x0, y0, x1, y1 = tiling._scale_rect(rects[rank], fw, fh, cols, rows)
            stamp = ((np.arange(rows)[:, None] * 7 + np.arange(cols * texel_bytes)[None, :] * 3) % 251).astype(np.uint8)
            expect = np.full((rows, cols * texel_bytes), 0xEE, np.uint8)
            hx0, hy0, hx1, hy1 = max(x0 - halo, 0), max(y0 - halo, 0), min(x1 + halo, cols), min(y1 + halo, rows)
            expect[hy0:hy1, hx0 * texel_bytes:hx1 * texel_bytes] = stamp[hy0:hy1, hx0 * texel_bytes:hx1 * texel_bytes]
            assert np.array_equal(img, expect), "tile %d, image at 1/%d resolution" % (rank, div)


def test_exchange_watchdog_names_the_exchange_that_never_completes():
    """a transport whose completion never arrives must not hang the frame silently. The watchdog of the native exchange
    (csrc/frontend/band_exchange.cpp) with a completion query that always answers "still running": nothing is reported before the deadline, the report
    after it names rank, exchange and phase; an entry that completes is dropped and never reported"""
    import ctypes as C
    import time
    from plainrenderer_amd import backend
    lib = C.CDLL(backend.LIB_PATH)
    QUERY = C.CFUNCTYPE(C.c_int, C.c_void_p)
    never = QUERY(lambda user: 0)
    state = {"done": False}
    later = QUERY(lambda user: 1 if state["done"] else 0)
    wd = C.c_void_p()
    assert lib.plrf_watchdog_create(C.c_uint32(150), C.byref(wd)) == 0
    buf = C.create_string_buffer(1024)
    assert lib.plrf_watchdog_arm(wd, C.c_int(3), C.c_int(2), C.c_int(0x100), never, None) == 0      # rank 3, temporally filtered GI, BEGIN
    assert lib.plrf_watchdog_arm(wd, C.c_int(3), C.c_int(4), C.c_int(0x100), later, None) == 0      # rank 3, resolved colour, BEGIN: completes in time
    assert lib.plrf_watchdog_poll(wd, buf, C.c_size_t(1024)) == 0 and buf.value == b""
    state["done"] = True
    time.sleep(0.3)
    assert lib.plrf_watchdog_poll(wd, buf, C.c_size_t(1024)) == 1
    msg = buf.value.decode()
    assert "rank 3" in msg and "exchange 2" in msg and "temporally filtered GI" in msg and "BEGIN" in msg and "deadline 150 ms" in msg, msg
    assert "exchange 4" not in msg
    assert lib.plrf_watchdog_destroy(wd) == 0
    # deadline 0 = off
    assert lib.plrf_watchdog_create(C.c_uint32(0), C.byref(wd)) == 0
    assert lib.plrf_watchdog_arm(wd, C.c_int(0), C.c_int(1), C.c_int(0x100), never, None) == 0
    time.sleep(0.05)
    assert lib.plrf_watchdog_poll(wd, buf, C.c_size_t(1024)) == 0
    assert lib.plrf_watchdog_destroy(wd) == 0


def _tile_rects(gx, gy):
    return tiling.tile_rects(W, H, gx, gy)


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["native", "python"])
@pytest.mark.parametrize("grid,half_res", [((2, 2), 1), ((2, 1), 1), ((2, 2), 0), ((4, 3), 1), ((1, 3), 1)])
def test_gpu_tiles_reproduce_the_full_frame_bit_exact(grid, half_res, transport):
    """BASELINE config 5's partition: the frame as 2 x 2 screen tiles (and 2 x 1; 4 x 3 tiles of 64 x 64 pixels and three whole-row rectangles, where tiles that do NOT
    touch exchange too: a plan reaches every rank within the halo), one backend per tile on this GPU, rectangles moved by the in-process transport.
    Halos as large as the image make every pass's inputs complete: the tiled frame must equal the unpartitioned one in
    every bit (exact kernel set), over three frames of temporal feedback - every column span, every valid-column range, every exchange rectangle, the per-tile
    pyramid and culling by tile columns, the histogram all-reduce of tile rectangles"""
    inputs = _make_inputs()
    full = _run_full(inputs, True, half_res)
    rects = _tile_rects(*grid)
    n = len(rects)
    big = max(W, H)
    halos = dict(band_gi_halo=big, band_gi_history_halo=big, band_post_halo=big, band_taa_history_halo=big, band_overlap_exchange=2 if half_res else 0)
    tiles = _run_bands(inputs, n, True, halos, half_res, rects=rects, transport=transport)
    mism = _compare(full, tiles, n, rects=rects)
    bad = {k: v for k, v in mism.items() if v != 0.0}
    assert not bad, bad
    B, E = 0x100, 0x200
    expected = [0, 1 | B, 1 | E, 2 | B, 2 | E, 3, 4 | B, 4 | E] if half_res else [0, 1, 2, 3, 4]
    if transport == "python":
        assert tiles[0]["calls"][:len(expected)] == expected


@pytest.mark.gpu
@pytest.mark.parametrize("grid,overlap", [((2, 2), None), ((1, 3), None), ((4, 3), None), ((2, 1), None), ((2, 2), 0), ((1, 3), 0)])
def test_gpu_request_lists_reproduce_the_full_frame_bit_exact(grid, overlap):
    """band_gi_halo = PLRF_HALO_REQUESTED (round 6): in front of the two spatial GI filter passes no halo is exchanged at all - every rank asks the
    owners for exactly the texels its disc samples land on (giSampleRequests passes -> bitmaps -> ExchangeGiRequests; then Y_SH, CoCg and depth of the marked texels at
    the two GI exchange points), through the native exchange over its in-process transport. With the other halos as large as the image the partitioned frame must
    equal the unpartitioned one in every bit (benchmarked kernel set: the requests are the fast filter kernel's own sample positions), over three frames of feedback."""
    inputs = _make_inputs()
    full = _run_full(inputs, False)
    rects = _tile_rects(*grid)
    n = len(rects)
    big = max(W, H)
    halos = dict(band_gi_halo=0xfffffffe, band_gi_history_halo=big, band_post_halo=big, band_taa_history_halo=big)
    if overlap is not None:  # band_overlap_exchange 0: one exchange callback and ONE filter execution per point (default: BEGIN / END around the filter's first phase)
        halos["band_overlap_exchange"] = overlap
    tiles = _run_bands(inputs, n, False, halos, rects=rects, transport="native")
    mism = _compare(full, tiles, n, rects=rects)
    bad = {k: v for k, v in mism.items() if v != 0.0}
    assert not bad, bad
    sent, received, groups = tiles[0]["exchange"][0]
    assert groups >= 5 and received > 0, tiles[0]["exchange"]


@pytest.mark.gpu
def test_gpu_tiles_fast_math_matches_full_frame_and_edges_first_changes_no_bit():
    """the benchmarked (PLR_MATH_FAST) kernel set through the 2 x 2 partition: whole-image halos -> the unpartitioned frame's bits; and with small halos the
    producers of exchanged images run the FRAME of the tile first (plr.h first_rows + first_cols, one launch, edge signal) - the same bits as the plain
    recording, the signal raised once per producer and frame"""
    inputs = _make_inputs()
    full = _run_full(inputs, False)
    rects = _tile_rects(2, 2)
    big = max(W, H)
    tiles = _run_bands(inputs, 4, False, dict(band_gi_halo=big, band_gi_history_halo=big, band_post_halo=big, band_taa_history_halo=big), rects=rects)
    mism = _compare(full, tiles, 4, rects=rects)
    bad = {k: v for k, v in mism.items() if v != 0.0}
    assert not bad, bad
    small = dict(band_gi_halo=8, band_gi_history_halo=8, band_post_halo=16, band_taa_history_halo=8)
    first = _run_bands(inputs, 4, False, dict(small, band_overlap_exchange=2), rects=rects)
    plain = _run_bands(inputs, 4, False, dict(small, band_overlap_exchange=0), rects=rects)
    for i in range(4):
        ptr, value, now = first[i]["edge_signal"]
        assert ptr is not None and value == 3 * N_FRAMES and now == value, (i, value, now)
        assert plain[i]["edge_signal"][1] == 0
        x0, y0, x1, y1 = rects[i]
        for f in range(N_FRAMES):
            for k in ("post", "color", "swap"):
                assert np.array_equal(first[i]["frames"][f][k][y0:y1, x0:x1], plain[i]["frames"][f][k][y0:y1, x0:x1]), "tile %d frame %d %s" % (i, f, k)


@pytest.mark.gpu
def test_gpu_tiles_of_realistic_size_with_default_halos(monkeypatch):
    """2 x 2 tiles of 512 x 576 pixels of a 1024 x 1152 frame with the DEFAULT halos, benchmarked kernel set, three frames - the tile version of the two-band test
    above: everything except the spatial GI filter is exact for these halos, the filter's disc gives samples beyond the halo weight 0"""
    import sys
    import parity
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "W", 1024)
    monkeypatch.setattr(mod, "H", 1152)
    inputs = _make_inputs()
    full = _run_full(inputs, False)
    rects = tiling.tile_rects(1024, 1152, 2, 2)
    tiles = _run_bands(inputs, 4, False, None, rects=rects)
    worst_within1, worst_mean = 1.0, 0.0
    for f in range(N_FRAMES):
        for i in range(4):
            x0, y0, x1, y1 = rects[i]
            fr, bf = full["frames"][f], tiles[i]["frames"][f]
            # the all-reduce gives every tile the same histogram; against the unpartitioned frame the handful of pixels of the previous frame that differ by a code
            # (the denoiser's halo deviation, below) may sit in a neighbouring bin
            assert np.array_equal(tiles[0]["frames"][f]["hist"], bf["hist"]) and int(bf["hist"].sum()) == 1024 * 1152, "histogram all-reduce: identical on every tile (frame %d)" % f
            assert int(np.abs(fr["hist"].astype(np.int64) - bf["hist"].astype(np.int64)).sum() // 2) <= 64, "histogram vs the unpartitioned frame (frame %d)" % f
            a_, b_ = np.ascontiguousarray(bf["post"][y0:y1, x0:x1]), np.ascontiguousarray(fr["post"][y0:y1, x0:x1])
            d = parity.r11g11b10_code_diff(a_.reshape(-1), b_.reshape(-1))
            within1 = float((d <= 1).all(axis=1).mean())
            a, b = pixfmt_unpack(a_), pixfmt_unpack(b_)
            mean_rel = float(np.abs(a - b).mean() / b.mean())
            print("TILES frame %d tile %d: pixels within one code of the unpartitioned frame %.5f, max code diff %d, mean rel err %.2e" % (f, i, within1, int(d.max()), mean_rel))
            worst_within1, worst_mean = min(worst_within1, within1), max(worst_mean, mean_rel)
    assert worst_within1 >= 0.995 and worst_mean <= 1e-3
    assert tiles[0]["fused"] >= 8, tiles[0]["fused"]  # tile mode keeps the fusions: per-tile pyramid + culling, upscale + shade, packed GI texels, apply + tonemap


@pytest.mark.gpu
@pytest.mark.parametrize("loopback", [False, True])
def test_gpu_native_exchange_packs_and_unpacks_rectangles(backend, loopback):
    """the native exchange's pack / unpack kernels and packed transport (csrc/frontend/band_exchange.cpp): rectangles of 4- and 8-byte texels, 16-byte aligned and
    not, travel pack -> ncclSend / ncclRecv to this rank itself (loopback: a device copy) -> unpack onto another place of the image"""
    from plainrenderer_amd.frame import FramePipeline
    w, h = 256, 128
    fp = FramePipeline(backend, w, h, shadow_map_res=128, brdf_lut_res=16, froxel_depth=8, max_sdf_instances=16, band_row_begin=0, band_row_end=h)
    try:
        fp.attach_rccl_rects(None if loopback else fp.rccl_unique_id(), 0, 1, w, h, [(0, 0, w, h)])
        info = fp.rccl_info()
        assert info["rccl_ranks"] == (0 if loopback else 1) and info["rccl_version"] > 0 and info["watchdog_ms"] == 2000, info
        rng = np.random.default_rng(5)
        for name, dtype, texel_bytes, rect, dst in (("post1", np.uint32, 4, (8, 4, 72, 36), (128, 64)), ("post1", np.uint32, 4, (3, 5, 14, 9), (100, 90)),
                                                   ("giFullResYSH", np.uint64, 8, (16, 8, 48, 24), (160, 96))):
            img = fp.image(name)
            data = rng.integers(0, np.iinfo(dtype).max, (h, w), dtype=dtype)
            backend.uploadImage(img, data)
            ptr, size = backend.imageDevicePointer(img, 0)
            assert size == w * h * texel_bytes
            fp.rccl_self_test_rect(ptr, w * texel_bytes, texel_bytes, rect, dst)
            got = backend.downloadImage(img, 0, dtype).reshape(h, w)
            expect = data.copy()
            x0, y0, x1, y1 = rect
            expect[dst[1]:dst[1] + (y1 - y0), dst[0]:dst[0] + (x1 - x0)] = data[y0:y1, x0:x1]
            assert np.array_equal(got, expect), (name, rect, dst)
    finally:
        fp.destroy()
