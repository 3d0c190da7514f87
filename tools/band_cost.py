"""Redundant compute of a partitioned frame, measured on ONE GPU: the N partitions of the 7680 x (1080 N) frame - row bands, or a grid of screen tiles (BASELINE
config 5: 2 x 2) - are timed one after the other, each with the halo it recomputes and the NATIVE exchange (csrc/frontend/band_exchange.cpp) over its in-process
transport: all ranks render the warm-up frames together (every exchange moves what a multi-GPU run moves), then the group is frozen and each rank is timed alone
with its neighbours' last frame in its halos (measure_all). Their times are summed and compared with the time of the unpartitioned frame.
    python tools/band_cost.py [N] [--tiles GXxGY] [--passes] [--balance] [--requested | --exact] [--loopback]
default N = 4: the 8K frame; --tiles 2x2: config 5's partition (GX * GY = N); --balance: rectangle sizes from measured times, as bench.py does;
--requested: band_gi_halo = PLRF_HALO_REQUESTED (request lists: byte-identical to the unpartitioned frame), --exact: PLRF_HALO_WHOLE_IMAGE, default: the bounded halo;
--loopback: the replay of rounds 3 - 5 (every partition on its own with the exchange in loopback: a tile's halos are its own texels, a band's halos nothing)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from plainrenderer_amd import RenderBackend, tiling
from plainrenderer_amd.frame import FramePipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
gx, gy = 1, n
if "--tiles" in sys.argv:
    gx, gy = (int(v) for v in sys.argv[sys.argv.index("--tiles") + 1].lower().split("x"))
    n = gx * gy
class A: pass
args = A(); args.grid = 16; args.sdf_res = 64; args.shadow_res = 2048; args.steps = 40; args.warmup = 5; args.profile_frames = 0
w, h = 7680, 1080 * n
kind = "tiles %dx%d" % (gx, gy) if gx > 1 else "%d bands" % n


MODE = {"halo": None, "exact": 0xffffffff, "requested": 0xfffffffe}[os.environ.get("PLR_BAND_COST_MODE", "requested" if "--requested" in sys.argv else ("exact" if "--exact" in sys.argv else "halo"))]
LOOPBACK = "--loopback" in sys.argv  # the old replay: every partition on its own, the exchange in loopback (a tile's halos are its own texels, a band's halos nothing)


def make_pipeline(rects, index, group=None):
    be = RenderBackend(w, h, device=0)
    kw, band = {}, None
    if index is not None:
        x0, y0, x1, y1 = rects[index]
        kw = dict(band_row_begin=y0, band_row_end=y1)
        if x0 != 0 or x1 != w:
            kw.update(band_col_begin=x0, band_col_end=x1)
        band = (y0, y1)
        if MODE is not None:
            kw.update(band_gi_halo=MODE)
    if index is not None and "PLR_BAND_COST_GI_HALO" in os.environ:  # experiment hook: trace texels of GI exchanged with each neighbour (tools/config5_series.sh: the exact mode)
        kw.update(band_gi_halo=int(os.environ["PLR_BAND_COST_GI_HALO"]))
    if index is not None and "PLR_BAND_COST_OVERLAP" in os.environ:  # experiment hook: band_overlap_exchange (plr_frame.h; 2 = edges first in one launch, the default)
        kw.update(band_overlap_exchange=int(os.environ["PLR_BAND_COST_OVERLAP"]))
    fp = FramePipeline(be, w, h, shadow_map_res=2048, **kw)
    if index is not None:
        if group is not None:
            fp.attach_local_rects(group, index, len(rects), w, h, rects)  # the native exchange over its in-process transport: real neighbours
        else:
            fp.attach_rccl_rects(None, index, len(rects), w, h, rects)  # loopback: the recording is the multi-GPU one, the local work of the exchange runs
    # (a halo beyond the default needs the inputs of rows beyond bench.input_halo: the whole frame's then)
    whole_inputs = MODE is not None or "PLR_BAND_COST_GI_HALO" in os.environ
    scene, cams, inputs = bench.build_scene(args, "cuda:0", w, h, None if whole_inputs else band)
    inputs.upload(fp)
    be.waitForGPUIdle()
    return be, fp, cams


def timed(be, fp, cams, index):
    t0 = time.perf_counter()
    for i in range(args.steps):
        fp.frame(cams[i + 6], 1 / 60, 0.5)
    be.waitForGPUIdle()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    if index is not None:
        sent, received, groups = fp.rccl_stats()
        print("    partition %d: %.1f MB sent, %.1f MB received per frame in %d point-to-point groups" % (index, sent / 1e6, received / 1e6, groups))
    be.setPassTiming(True)
    acc = {}
    for i in range(8):
        fp.frame(cams[i + 20], 1 / 60, 0.5)
        for name, t in be.getRenderpassTimings():
            acc[name] = acc.get(name, 0.0) + t / 8
    be.setPassTiming(False)
    return ms, acc


def measure(rects, index):
    """one partition (index None: the unpartitioned frame) on its own, the exchange in loopback"""
    be, fp, cams = make_pipeline(rects, index)
    for i in range(args.warmup):
        fp.frame(cams[i + 1], 1 / 60, 0.5)
    be.waitForGPUIdle()
    out = timed(be, fp, cams, index)
    fp.destroy()
    be.shutdown()
    return out


def measure_all(rects):
    """Every partition timed ALONE on the GPU with REAL neighbour data in its halos (round 6): all ranks live in this process (a thread, backend,
    pipeline and native exchange each, over the in-process transport), render the warm-up frames together - every exchange moves what a multi-GPU run moves -, then the
    group is FROZEN: a rank no longer waits for its peers and copies from what they posted last (their buffers keep their last frame), and the ranks are timed one
    after the other. -> [(ms, pass times)] per rank"""
    if LOOPBACK:
        return [measure(rects, i) for i in range(len(rects))]
    import threading
    from plainrenderer_amd.frame import LocalExchangeGroup
    nr = len(rects)
    group = LocalExchangeGroup(nr)
    warm, go, done = threading.Barrier(nr + 1), [threading.Event() for _ in range(nr)], [threading.Event() for _ in range(nr)]
    results, errors = [None] * nr, []

    def rank(i):
        try:
            be, fp, cams = make_pipeline(rects, i, group)
            for f in range(args.warmup):
                fp.frame(cams[f + 1], 1 / 60, 0.5)
            be.waitForGPUIdle()
            warm.wait()
            go[i].wait()
            results[i] = timed(be, fp, cams, i)
            done[i].set()
            go[i].clear(); go[i].wait()  # the pipelines stay alive until every rank has been timed: the others copy from this one's buffers
            fp.destroy()
            be.shutdown()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            group.abort()
            try:
                warm.abort()
            except Exception:  # noqa: BLE001
                pass
            done[i].set()
            raise

    threads = [threading.Thread(target=rank, args=(i,)) for i in range(nr)]
    for t in threads:
        t.start()
    try:
        warm.wait(timeout=900)
    except threading.BrokenBarrierError:
        raise SystemExit("band_cost: a rank failed during the warm-up frames: %s" % (errors[:1],))
    group.freeze(True)
    for i in range(nr):
        go[i].set()
        done[i].wait(timeout=900)
        if errors:
            raise SystemExit("band_cost: rank %d failed: %s" % (i, errors[0]))
    for i in range(nr):
        go[i].set()
    for t in threads:
        t.join(timeout=300)
    group.destroy()
    return results


measure(None, None)  # (the first measurement of a process reads up to 30 % high on some boxes: taken twice, the second is reported)
full, full_passes = measure(None, None)
print("unpartitioned %dx%d frame: %.3f ms" % (w, h, full))
cols, rows = tiling.equal_bounds(w, gx), tiling.equal_bounds(h, gy)
rects = tiling.tile_rects(w, h, gx, gy, cols, rows)
if "--balance" in sys.argv:
    # what bench.py --gpus N does before its timed region (static load balancing from measured times), here with the partitions one after the other
    best, seen = None, []
    for it in range(4):  # up to four rounds; the partition used is the best one MEASURED (bench.calibrate_partition)
        times = [r[0] for r in measure_all(rects)]
        print("partition %s: times %s ms, slowest / mean = %.3f" % (rects, ["%.3f" % t for t in times], max(times) / (sum(times) / n)))
        seen.append((list(cols), list(rows)))
        if best is None or max(times) < best[0]:
            best = (max(times), list(cols), list(rows))
        new_cols, new_rows = tiling.balanced_tile_bounds(w, h, gx, gy, cols, rows, times, min_size=512)
        if (list(new_cols), list(new_rows)) in seen:
            break
        cols, rows = new_cols, new_rows
        rects = tiling.tile_rects(w, h, gx, gy, cols, rows)
    cols, rows = best[1], best[2]
    rects = tiling.tile_rects(w, h, gx, gy, cols, rows)
total, band_passes, slowest = 0.0, {}, 0.0
final = measure_all(rects)
for i in range(n):
    ms, acc = final[i]
    total += ms
    slowest = max(slowest, ms)
    for k, v in acc.items():
        band_passes[k] = band_passes.get(k, 0.0) + v
    print("partition %d of %d (%s, columns %d..%d, rows %d..%d): %.3f ms" % (i, n, kind, rects[i][0], rects[i][2], rects[i][1], rects[i][3], ms))
def group(d):  # fused launches of the unpartitioned frame count towards their passes' group
    g = {}
    for k, v in d.items():
        key = "GI trace + filters" if ("Indirect" in k) else ("bloom" if "loom" in k or "Tonemap" in k else ("exposure" if "istogram" in k or "expose" in k else k.split(" + ")[0]))
        g[key] = g.get(key, 0.0) + v
    return g
if "--passes" in sys.argv:
    for k in sorted(set(full_passes) | set(band_passes), key=lambda k: -band_passes.get(k, 0.0)):
        print("    %-60s partitions %.4f ms (%.4f each), unpartitioned %.4f ms" % (k[:60], band_passes.get(k, 0.0), band_passes.get(k, 0.0) / n, full_passes.get(k, 0.0)))
gf, gb = group(full_passes), group(band_passes)
for k in sorted(gb, key=lambda k: -gb[k]):
    print("  %-34s partitions %.3f ms, unpartitioned %.3f ms (%+.1f %%)" % (k, gb[k], gf.get(k, 0.0), 100.0 * (gb[k] / gf[k] - 1.0) if gf.get(k) else 0.0))
print("%s: sum of the partitions %.3f ms = %.3f x the unpartitioned frame: %.1f %% redundant compute; slowest partition %.3f ms -> at most %.2fx on %d GPUs before any exchange wait" % (
    kind, total, total / full, 100.0 * (total / full - 1.0), slowest, full / slowest, n))
