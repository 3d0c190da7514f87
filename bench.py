#!/usr/bin/env python
"""Headline benchmark: frames/s of the full GI + shade + post frame at 3840x2160 on MI355X (BASELINE.json).

One step = one frame of the hot path recorded by the C++ FramePipeline (reference pass order: histogram x3, pre-expose, HiZ,
depth downscale, SDF frustum+tile culling, diffuse trace, spatial/temporal/spatial denoise, upscale, deferred shade, TAA,
bloom x11, tonemap) on synthetic inputs that are resident in HBM before the timed region starts.

  python bench.py --gpus N --steps K --warmup W

N > 1: ONE frame of N x the 4K pixel count (7680 x 1080*N; N = 4 is the 7680x4320 frame of BASELINE config 5), partitioned into N
rectangles, one rank per GPU - N row bands (the default: the faster partition in the single-GPU replay of both, profiles/r05_tile_vs_band.txt) or 2 x N/2 SCREEN
TILES (--partition tiles; N = 4: config 5's 2 x 2) - halos exchanged over RCCL point-to-point (C++ host, csrc/frontend/band_exchange.cpp) plus one 512-byte
histogram all-reduce; weak scaling (every GPU keeps one 4K frame's worth of pixels), and the JSON line also carries the STRONG-scaling figure the
target is written in: the time of the same frame unpartitioned on one GPU (measured on rank 0 in the same run) over the N-GPU time. The rectangles'
sizes are balanced before the timed region from measured render times (up to four calibration rounds, the best measured partition is kept; `band_partition` in the JSON line; --no-balance keeps equal sizes). When WORLD_SIZE is not set, `--gpus N` spawns the N
ranks itself (python -m torch.distributed.run, rendezvous on 127.0.0.1); under torchrun WORLD_SIZE must equal --gpus. A band frame that
cannot run fails the benchmark (non-zero exit) unless --allow-replicas is given.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def algorithmic_bytes(w, h, n_instances, sdf_res, shadow_res, brdf_res, froxel_depth):
    """Compulsory HBM bytes per pass launch: every input read once, every output written once."""
    N = w * h
    M = (w // 2) * (h // 2)
    tiles = math.ceil(w / 32) * math.ceil(h / 32)
    sdf = n_instances * sdf_res ** 3 * 2
    lut = brdf_res * brdf_res * 8 + 3 * shadow_res * shadow_res * 2 + math.ceil(w / 8) * math.ceil(h / 8) * froxel_depth * 8
    b = {
        "Histogram per tile": 4 * N + tiles * 512,
        "Histogram reset": 512,
        "Histogram combine tiles": tiles * 512 + 512,
        "Pre-expose lights": 512 + 20 + 4,
        "Depth min/max pyramid": 4 * N + 8 * (N / 4) * 4.0 / 3.0,
        "Depth downscale": 6 * M,
        "SDF camera frustum culling": n_instances * 36,
        "SDF camera tile culling": n_instances * 36 + tiles * 404 / 4,
        "Indirect diffuse SDF trace": 20 * M + sdf + (tiles / 4) * 404,
        "Indirect diffuse spatial filter": 30 * M,
        "Indirect diffuse spatial filter (texel packing)": 30 * M,  # pre-pass of the fast kernel set: reads the 14 B/px inputs, writes 16 B/px packed texels
        "Indirect diffuse temporal filter": 56 * M,
        "Indirect lighting upscale": 14 * M + 16 * N,
        "Forward shading (deferred)": 32 * N + lut,
        "Temporal filtering": 24 * N,
        "Apply bloom": 12 * N,
        "Tonemap": 8 * N,
    }
    for m in range(1, 6):  # down: read mip m-1, write mip m; up (target t): read down t+1, up t+1, write t
        b["Bloom downsample mip %d" % m] = 4 * N / 4 ** (m - 1) + 4 * N / 4 ** m
    for t in range(4, -1, -1):
        b["Bloom Upsample mip %d" % t] = 4 * N / 4 ** (t + 1) * (1 if t == 4 else 2) + 4 * N / 4 ** t
    frame = 157.3 * N + sdf + lut
    return b, frame


def input_halo(height):
    """full-res rows of G-buffer a band needs beyond its own: 2 * (giHalo + giHistoryHalo) + 16 for the half-res depth (giHalo = 64 trace rows per 2160
    rows of frame height, plrf_default_settings), one more 64-row tile for the per-tile depth pyramid; rounded up to 128"""
    gi = 64 * ((height + 2159) // 2160)
    return ((2 * (gi + 16) + 16 + 64 + 127) // 128) * 128


PASS_KERNEL = {  # pass label -> kernel name prefixes in the rocprofv3 summaries under profiles/ (a pass may launch more than one kernel)
    "Indirect diffuse spatial filter": ["plr::spatialFilter"], "Indirect diffuse spatial filter (texel packing)": ["plr::spatialPack"], "Forward shading (deferred)": ["plr::fastshade::deferredShading"],
    "Temporal filtering": ["plr::fasttaa::temporalFilter"], "Indirect diffuse SDF trace": ["plr::fasttrace::sdfDiffuseTrace"],
    "Indirect lighting upscale": ["plr::faststream::indirectLightUpscale"], "Indirect lighting upscale + Forward shading (deferred)": ["plr::fastshade::upscaleAndShade"], "Indirect diffuse temporal filter": ["plr::faststream::temporalGiFilter"],
    "Depth min/max pyramid": ["plr::hizBase", "plr::hizTail"], "Tonemap": ["plr::faststream::tonemapping"], "Apply bloom": ["plr::faststream::applyBloom"],
    "Histogram per tile": ["plr::fasthist::histogramPerTile"], "Apply bloom + Tonemap": ["plr::faststream::applyBloomTonemap"],
    "Histogram per tile + Histogram reset + Histogram combine tiles + Pre-expose lights + Depth min/max pyramid + Depth downscale": ["plr::fasthist::histogramAndPyramid", "plr::exposureChainAndPyramidTail"],
    "Histogram per tile + Histogram reset + Histogram combine tiles + Pre-expose lights + Depth min/max pyramid + Depth downscale + SDF camera frustum culling + SDF camera tile culling":
        ["plr::fasthist::histogramAndPyramid", "plr::exposureChainAndPyramidTail"],
    "SDF camera frustum culling + SDF camera tile culling": ["plr::frustumAndTileCulling"], "Bloom Upsample mip 0": ["plr::fastbloom::bloomUpsampleQuad"],
}


def kernel_source_digest():
    """sha1 over the kernel sources (plainrenderer_amd/csrc): stamps a PMC summary with the build it was measured on"""
    import hashlib
    h = hashlib.sha1()
    base = os.path.join(ROOT, "plainrenderer_amd", "csrc")
    for d, _, files in sorted(os.walk(base)):
        if os.path.basename(d).startswith("_obj"):
            continue
        for f in sorted(files):
            if f.endswith((".hip", ".h", ".cpp")):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def pmc_traffic(pass_name):
    """HBM-side bytes per launch of the pass (all its kernels) from the newest committed PMC summary (profiles/*_pmc_hbm.csv: separate
    FETCH_SIZE / WRITE_SIZE passes of this same command at the default workload, gfx950 x2 fetch correction applied)."""
    import glob
    prefixes = PASS_KERNEL.get(pass_name)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.csv")))
    if not prefixes or not files:
        return None, None
    total, found, digest = 0, False, None
    for line in open(files[-1]):
        if line.startswith("# kernel source digest:"):
            digest = line.split(":", 1)[1].strip()
        if any(line.startswith(p) for p in prefixes):
            # mean bytes per dispatch of that kernel; a kernel launched k times per frame by several passes (bloom levels) is not attributed here
            total += int(line.strip().split(",")[-1])
            found = True
    # the summary is only as good as the build it was taken on: say whether the kernels have changed since
    src = os.path.basename(files[-1]) + (" (kernel sources unchanged since)" if digest == kernel_source_digest() else " (kernel sources CHANGED since: stale)" if digest else "")
    return (total, src) if found else (None, None)


def pmc_counters(pass_name):
    """Per-launch SQ / TCP counters of the pass's kernel(s) from the newest committed profiles/*_sq_counters.csv (tools/profile_round.sh: rocprofv3 --pmc
    passes of this same command at the default workload, means per dispatch by tools/pmc_summary.py) -> ({counter: value}, source note) or (None, None)"""
    import glob
    prefixes = PASS_KERNEL.get(pass_name)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.csv")))
    if not prefixes or not files:
        return None, None
    digest, header, acc = None, None, {}
    for line in open(files[-1]):
        line = line.strip()
        if line.startswith("# kernel source digest:"):
            digest = line.split(":", 1)[1].strip()
        elif line.startswith("kernel,"):
            header = line.split(",")
        elif header and any(line.startswith(p) for p in prefixes):
            cols = line.split(",")
            for k, v in zip(header[2:], cols[2:]):
                try:
                    acc[k] = acc.get(k, 0.0) + float(v)
                except ValueError:
                    pass
    src = os.path.basename(files[-1]) + (" (kernel sources unchanged since)" if digest == kernel_source_digest() else " (kernel sources CHANGED since: stale)" if digest else "")
    return (acc, src) if acc else (None, None)


SIMDS, CUS, SHADER_CLOCK_GHZ = 1024, 256, 2.4  # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, 2.4 GHz nominal
DENSE_VALU_CLOCK_GHZ = 2.0  # profiles/r05_valu_rates.txt, column "clock GHz": a chip-wide dense full-rate VALU stream sustains 1.85 - 2.0 GHz (DVFS)


def _newest_profile(suffix):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)))
    return files[-1] if files else None


def _norm_kernel(name):
    """kernel names differ between the rocprofv3 summaries (template arguments separated by ', ' or '; '): one spelling"""
    return name.replace("; ", ", ").strip().strip('"')


def frame_valu_floor():
    """The whole frame's VALU issue floor from the newest committed counter set (profiles/*_sq_counters.csv + *_kernel_stats.csv of the same round):
    sum over the frame's launches of SQ_INSTS_VALU x 2 cycles / 1024 SIMDs / clock - what the frame would take if every SIMD issued one VALU instruction every
    2 cycles and nothing else ever waited. The clock is MEASURED per kernel, GRBM_GUI_ACTIVE / 8 XCDs over the kernel's traced duration, capped at the nominal
    2.4 GHz (the counter also ticks a little before and after the traced interval); the same floor at the 2.0 GHz a dense VALU stream sustains is given beside it.
    -> dict or None"""
    cfile, kfile = _newest_profile("_sq_counters.csv"), _newest_profile("_kernel_stats.csv")
    if not cfile or not kfile or os.path.basename(cfile).split("_")[0] != os.path.basename(kfile).split("_")[0]:
        return None
    dur = {}
    for line in open(kfile):
        if line.startswith("kernel,") or line.startswith("#"):
            continue
        # the kernel name contains commas: the last six fields are the numbers
        parts = line.rstrip("\n").rsplit(",", 6)
        if len(parts) == 7:
            dur[_norm_kernel(parts[0])] = float(parts[2])
    header, rows, digest = None, [], None
    for line in open(cfile):
        line = line.strip()
        if line.startswith("# kernel source digest:"):
            digest = line.split(":", 1)[1].strip()
        elif line.startswith("kernel,"):
            header = line.split(",")
        elif header and line and not line.startswith("#"):
            cols = line.split(",")
            rows.append(dict(zip(header, cols)))
    if not rows:
        return None
    per_frame = None
    for r in rows:
        if "histogramAndPyramid" in r["kernel"]:
            per_frame = float(r["dispatches"])
    if not per_frame:
        return None
    floor_measured = floor_nominal = 0.0
    clocks, kernels = [], []
    for r in rows:
        launches = float(r["dispatches"]) / per_frame
        if launches < 0.5:  # set-up kernels (LUT bake, tables): not part of a frame
            continue
        valu = float(r.get("SQ_INSTS_VALU", 0) or 0)
        d_us = dur.get(_norm_kernel(r["kernel"]))
        grbm = float(r.get("GRBM_GUI_ACTIVE", 0) or 0) / 8.0
        clock = min(grbm / (d_us * 1e3), SHADER_CLOCK_GHZ) if (d_us and grbm) else SHADER_CLOCK_GHZ
        floor_measured += launches * valu * 2.0 / SIMDS / (clock * 1e9) * 1e3
        floor_nominal += launches * valu * 2.0 / SIMDS / (SHADER_CLOCK_GHZ * 1e9) * 1e3
        if d_us and grbm:
            clocks.append((launches * valu, clock))
        kernels.append(_norm_kernel(r["kernel"]).split("<")[0].split("::")[-1])
    wsum = sum(w for w, _ in clocks)
    clock_mean = sum(w * c for w, c in clocks) / wsum if wsum else SHADER_CLOCK_GHZ
    src = os.path.basename(cfile) + " + " + os.path.basename(kfile) + (" (kernel sources unchanged since)" if digest == kernel_source_digest() else " (kernel sources CHANGED since: stale)" if digest else "")
    return {"frame_valu_floor_ms": round(floor_measured, 4), "clock_GHz_measured": round(clock_mean, 3),
            "clock_source": "GRBM_GUI_ACTIVE / 8 XCDs over the traced kernel duration, VALU-weighted mean over the frame's kernels, capped at the nominal 2.4 GHz",
            "frame_valu_floor_ms_at_2.0GHz_dense_valu_clock": round(floor_nominal * SHADER_CLOCK_GHZ / DENSE_VALU_CLOCK_GHZ, 4),
            "cycles_per_valu_instruction_at_peak": 2, "simds": SIMDS, "kernels": len(kernels), "source": src}


def build_scene(args, device, w, h, band=None):
    """band = (row_begin, row_end) of this rank in the w x h frame, or None for the whole frame"""
    from plainrenderer_amd import synth
    from plainrenderer_amd.frame import SyntheticInputs
    from plainrenderer_amd.scene import Camera
    n_cams = args.steps + args.warmup + args.profile_frames + 28
    if getattr(args, "scene", "default") == "dense":
        # --scene dense: the instances packed one metre apart (they overlap: 0.8 - 2.5 m half extents), the camera looking along the field - culling
        # tiles carry tens of instances up to the list's cap of 100 (sdfCameraTileCulling.comp:42-99), the generator of tests/test_sdfgi.py dense_scene at the bench's size
        scene = synth.SynthScene(grid=args.grid, cell=1.0, seed_id=301, device=device)
        cams = [Camera.look((args.grid * 0.5 + 0.002 * i, -5.0, -6.0 + 0.004 * i), (0.0, 0.35, 1.0), aspect=w / h) for i in range(n_cams)]
    else:
        scene = synth.SynthScene(grid=args.grid, cell=8.0, seed_id=700, device=device)
        span = args.grid * 8.0
        x0 = span * 0.35
        cams = [Camera.look((x0 + 0.002 * i, -9.0, -10.0 + 0.004 * i), (0.02, 0.17, 1.0), aspect=w / h) for i in range(n_cams)]
    rows = depth_range = None
    if band is not None:
        # every band must fit the same shadow cascades: depth range of the whole frame from a 1/8-resolution G-buffer (same on all ranks)
        coarse = scene.gbuffer(cams[1], w // 8, h // 8)["depth"]
        vis = coarse[coarse > 0].astype(np.float64)
        n, f = cams[1].near, cams[1].far
        lin = n * f / (f + (1.0 - vis) * (n - f)) if vis.size else np.array([1.0, 50.0])
        depth_range = (float(lin.min()), float(lin.max()))
        rows = (max(band[0] - input_halo(h), 0), min(band[1] + input_halo(h), h))
    inputs = SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=args.sdf_res, shadow_res=args.shadow_res, froxel_depth=64, sun_direction=(0.35, -0.8, 0.45),
                             rows=rows, depth_range=depth_range)
    return scene, cams, inputs


def cpu_baseline(args, cores, device):
    """oracle frames (scalar C++ restatement, rows split over `cores` std::threads) on a quarter-area (1920x1080) version of the
    same workload; reported as 4K-equivalent frames/s (time scaled by the pixel ratio). Inputs are generated on `device`."""
    import pyoracle
    from oracle_frame import OracleFrame
    from plainrenderer_amd import synth
    from plainrenderer_amd.frame import PlrfSettings, SyntheticInputs
    from plainrenderer_amd.scene import Camera, GlobalShaderInfo, taa_jitter_pixels, taa_resolve_weights
    scale = 2
    w, h = args.width // scale, args.height // scale
    pyoracle.set_threads(cores)
    scene = synth.SynthScene(grid=args.grid, cell=8.0, seed_id=700, device=device)
    span = args.grid * 8.0
    cams = [Camera.look((span * 0.35 + 0.002 * i, -9.0, -10.0 + 0.004 * i), (0.02, 0.17, 1.0), aspect=w / h) for i in range(5)]  # the GPU leg's camera path
    inputs = SyntheticInputs(scene, cams[1], cams[0], w, h, sdf_res=args.sdf_res, shadow_res=args.shadow_res, froxel_depth=64, sun_direction=(0.35, -0.8, 0.45))
    inputs.volume_indices = list(range(len(inputs.volumes)))
    inputs.instance_bytes_patched = inputs.instance_bytes
    s = PlrfSettings()
    s.taa_enabled = s.taa_use_clipping = s.taa_use_motion_vector_dilation = s.taa_filter_use_tonemapping = 1
    s.taa_history_sampling_tech = 4
    s.bloom_enabled, s.bloom_strength, s.bloom_radius = 1, 0.05, 1.5
    s.sdf_half_res_trace, s.sdf_strict_influence_radius_cutoff, s.sdf_trace_influence_radius = 1, 1, 5.0
    s.diffuse_brdf, s.direct_multiscatter, s.indirect_lighting_tech, s.use_geometry_aa, s.sun_shadow_cascade_count = 2, 0, 0, 1, 3
    s.run_exposure = s.run_hiz = s.run_gi = s.run_shading = s.run_taa = s.run_bloom = s.run_tonemap = 1
    ora = OracleFrame(inputs, w, h, 512, s)
    n_vol = len(inputs.volumes)
    times = []
    for f in range(4):
        g = GlobalShaderInfo(frameIndex=f, sunDirection=(*inputs.sun.tolist(), 0.0), time=0.5 + f / 60.0, deltaTime=1 / 60.0)
        g.noiseTextureIndices = (n_vol, n_vol + 1, n_vol + 2, n_vol + 3)
        cams[f + 1].fill_global(g, w, h)
        fp_, fn_ = cams[f].frustum_points_normals()
        frustum = np.concatenate([fp_.reshape(-1), fn_.reshape(-1)]).astype(np.float32).tobytes()
        t0 = time.perf_counter()
        ora.frame(g.pack(), taa_resolve_weights(taa_jitter_pixels((f + 1) % 8)), frustum, 5.0)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))  # frame 0 bakes the BRDF LUT; three timed frames
    return {"value": 1.0 / (t * scale * scale), "unit": "frames/s (3840x2160-equivalent)", "cores": cores, "kind": "port",
            "sample": "full frame at %dx%d (1/%d of the pixels; %d instances x %d^3 SDF, %d^2 shadow cascades, 64 froxel slices, 512^2 BRDF LUT as on the GPU), "
                      "median of 3 timed frames (after the one that bakes the BRDF LUT) = %.2f s each, scaled by the pixel ratio" % (w, h, scale * scale, args.grid ** 2, args.sdf_res, args.shadow_res, t)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600, help="timed frames (default: ~0.6 s of GPU time at 4K)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--grid", type=int, default=16, help="SDF instances = grid^2 (16 -> 256)")
    ap.add_argument("--sdf-res", type=int, default=64)
    ap.add_argument("--shadow-res", type=int, default=2048)
    ap.add_argument("--profile-frames", type=int, default=20, help="extra frames with per-pass hipEvent timing for the roofline object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pass-table", action="store_true", help="print the per-pass table to stderr")
    ap.add_argument("--producers", action="store_true", help="NOT the headline workload: also run the input producers as compute passes every frame (sun light matrices, "
                    "the three sky LUTs, the four froxel passes: SURVEY 8 f3) instead of reading uploaded LUTs / matrices; single GPU only")
    ap.add_argument("--scene", choices=["default", "dense"], default="default", help="NOT the headline workload: dense = the same 256 instances packed one metre apart, the camera looking "
                    "along the field - culling tiles carry tens of instances, up to the cap of 100 (the trace where the shader is stressed); the JSON line carries trace samples/s and the tile counts")
    ap.add_argument("--exact", action="store_true", help="diagnostic: run the bit-exact kernel set (PLR_MATH_EXACT) instead of the default fast set")
    ap.add_argument("--force-bands", action="store_true", help="diagnostic: run the N=1 frame through the band path (one band, RCCL group of size 1)")
    ap.add_argument("--python-exchange", action="store_true", help="diagnostic: drive the halo exchange from Python (torch.distributed) instead of the C++ host's RCCL exchange")
    ap.add_argument("--allow-replicas", action="store_true", help="N > 1 only: if the band frame cannot run, fall back to N independent 4K replicas (said so in the JSON line) instead of failing")
    ap.add_argument("--no-balance", action="store_true", help="N > 1: keep the equal partition instead of balancing the rectangles' sizes from measured times")
    ap.add_argument("--partition", choices=["auto", "tiles", "bands"], default="auto", help="N > 1: screen tiles (2 x N/2 grid; N = 4: BASELINE config 5's 2 x 2) or row bands; "
                    "auto (N even): BOTH are timed in this run (a short run each), the faster one is the headline, the other is reported under alt_partition")
    ap.add_argument("--gi-exchange", choices=["requested", "exact", "halo"], default="requested", help="N > 1: what the spatial GI filters get from the other ranks. requested (default): "
                    "request lists, band_gi_halo = PLRF_HALO_REQUESTED - every rank asks the owners for exactly the texels its samples land on; the partitioned frame EQUALS the unpartitioned one "
                    "bit for bit (tests/test_config5_8k.py). exact: every GI texel to every rank (PLRF_HALO_WHOLE_IMAGE), also bit-identical, 10 x the bytes. halo: 64 trace rows per 2160 frame "
                    "rows - NOT the same frame: samples beyond the halo get weight 0 and the deviation grows with the frame count (profiles/r05_config5_series.txt)")
    ap.add_argument("--exact-partition", action="store_true", help="the same as --gi-exchange exact")
    ap.add_argument("--no-strong-scaling", action="store_true", help="N > 1: skip the single-GPU run of the same frame (rank 0) the strong-scaling figure is taken against")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when --gpus N spawns its own ranks (0: derived from the pid)")
    args = ap.parse_args()
    if args.exact_partition:
        args.gi_exchange = "exact"
    # the byte-identical modes get the G-buffer inputs of the WHOLE frame on every rank (whole-image halo: the denoiser weighs samples anywhere in the frame; request lists:
    # kept the same so that both parity-true modes are measured on the same inputs - the requested texels arrive with their depth, so the rectangle + halo would do)
    args.whole_frame_inputs = args.gi_exchange != "halo"

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launched as `python bench.py --gpus N`: become the launcher of N ranks (one per GPU) and relay rank 0's JSON line
        import subprocess
        port = args.master_port or (20000 + os.getpid() % 20000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))

    # stdout carries exactly ONE line, the JSON result of rank 0: everything else a library may print there (RCCL's version banner is written to
    # the C stdout and flushed at exit, i.e. after the JSON line) is sent to stderr by pointing file descriptor 1 at it for the life of the process
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d; a scaling run must measure what it was asked for" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if torch.cuda.device_count() < (world if "LOCAL_RANK" in os.environ else 1):
        raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) visible on this node" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_bands:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    device = "cuda:%d" % local_rank

    from plainrenderer_amd import RenderBackend, tiling
    from plainrenderer_amd.frame import FramePipeline

    band_partition = {"bounds": None, "calibration": []}
    # auto = the measured winner of the single-GPU replay of both partitions from one build (profiles/r05_tile_vs_band.txt: slowest of four row bands 0.88 ms, slowest of
    # 2 x 2 tiles 0.91 ms of the 2.72 ms 8K frame): row bands. Config 5's own 2 x 2 tiles: --partition tiles. Neither has been measured on more than one GPU.
    if args.partition == "tiles" and world > 1 and world % 2:
        raise SystemExit("bench.py: --partition tiles needs an even number of GPUs (2 x N/2 grid)")
    geometry = {"tiles": world > 1 and world % 2 == 0 and args.partition == "tiles"}  # the geometry make() builds; "auto" tries both below

    calibrated = {}

    def grid():
        return (2, world // 2) if geometry["tiles"] else (1, world)

    def band_frame_size():
        # N GPUs render ONE frame of N x the pixels, partitioned by rows: 7680 x (1080 * N) (N = 4: the 8K frame of BASELINE config 5).
        # Every band is 7680 x ~1080 = one 4K frame's worth of pixels per GPU (weak scaling).
        return (2 * args.width, (args.height // 2) * world) if world > 1 else (args.width, args.height)

    def rect_settings(rect, w_):
        """FramePipeline settings of one rectangle of the partition"""
        kw = dict(band_row_begin=rect[1], band_row_end=rect[3])
        if rect[0] != 0 or rect[2] != w_:
            kw.update(band_col_begin=rect[0], band_col_end=rect[2])
        if args.gi_exchange == "exact":
            kw.update(band_gi_halo=0xffffffff)  # PLRF_HALO_WHOLE_IMAGE
        elif args.gi_exchange == "requested":
            kw.update(band_gi_halo=0xfffffffe)  # PLRF_HALO_REQUESTED
        return kw

    def calibrate_partition():
        """Static load balancing: the rectangles are rendered with an exchange that moves nothing (compute only; with the real exchange every rank would
        show the slowest rank's time), the per-rank times are gathered and the row / column boundaries re-cut so that every rectangle costs the same
        (tiling.balanced_tile_bounds). Two rounds; the same partition on every rank."""
        w_, h_ = band_frame_size()
        grid_x, grid_y = grid()
        cols, rows = tiling.equal_bounds(w_, grid_x), tiling.equal_bounds(h_, grid_y)
        best = None  # (slowest time, cols, rows) of the partitions MEASURED so far: the cost density inside a rectangle is not flat (sky above ground), so a re-cut
        # overshoots or undershoots; up to four rounds, and the partition returned is the best one that was actually timed, not the last proposal
        for _ in range(4):
            rects_ = tiling.tile_rects(w_, h_, grid_x, grid_y, cols, rows)
            b0, b1 = rects_[rank][1], rects_[rank][3]
            be_ = RenderBackend(w_, h_, device=local_rank)
            fp_ = FramePipeline(be_, w_, h_, shadow_map_res=args.shadow_res, **rect_settings(rects_[rank], w_))
            fp_.set_exchange_callback(lambda exchange_id, stream: None)
            _, cams_, inputs_ = build_scene(args, device, w_, h_, None if args.whole_frame_inputs else (b0, b1))
            inputs_.upload(fp_)
            for i in range(4):
                fp_.frame(cams_[i + 1], 1.0 / 60.0, 0.5)
            be_.waitForGPUIdle()
            tc = time.perf_counter()
            for i in range(12):
                fp_.frame(cams_[i + 5], 1.0 / 60.0, 0.5)
            be_.waitForGPUIdle()
            mine = (time.perf_counter() - tc) / 12.0
            fp_.destroy()
            be_.shutdown()
            del inputs_
            torch.cuda.empty_cache()
            times = torch.zeros(world, dtype=torch.float64, device=device)
            times[rank] = mine
            if world > 1:
                dist.all_reduce(times, op=dist.ReduceOp.SUM)
            times = [float(v) for v in times.cpu().tolist()]
            band_partition["calibration"].append({"col_bounds": list(cols), "row_bounds": list(rows), "partition_ms": [round(t * 1e3, 4) for t in times]})
            if best is None or max(times) < best[0]:
                best = (max(times), list(cols), list(rows))
            new = tiling.balanced_tile_bounds(w_, h_, grid_x, grid_y, cols, rows, times, min_size=512)  # every rectangle stays larger than the widest halo (224)
            if new == (cols, rows) or any(new == (c["col_bounds"], c["row_bounds"]) for c in band_partition["calibration"]):
                break
            cols, rows = new
        return best[1], best[2]

    def make(mode):
        """mode 'single': the 4K frame on this GPU; 'bands': one band of the N x larger frame; -> (be, fp, scene-tuple, w, h, band)"""
        w_, h_ = args.width, args.height
        band_ = None
        bounds_ = None
        rects_ = None
        if mode == "bands":
            w_, h_ = band_frame_size()
            grid_x, grid_y = grid()
            band_partition["calibration"] = []
            if args.no_balance:
                cols_, rows_ = tiling.equal_bounds(w_, grid_x), tiling.equal_bounds(h_, grid_y)
            else:  # (calibrated once per geometry: the short run that chooses the geometry and the headline run use the same partition)
                if geometry["tiles"] not in calibrated:
                    calibrated[geometry["tiles"]] = (calibrate_partition(), list(band_partition["calibration"]))
                (cols_, rows_), band_partition["calibration"] = calibrated[geometry["tiles"]]
            rects_ = tiling.tile_rects(w_, h_, grid_x, grid_y, cols_, rows_)
            band_ = (rects_[rank][1], rects_[rank][3])
            band_partition["bounds"] = rows_
            band_partition.update(kind="tiles %dx%d" % (grid_x, grid_y) if grid_x > 1 else "row bands", col_bounds=cols_, row_bounds=rows_, rects=[list(r) for r in rects_],
                                  gi_halo={"requested": "request lists (PLRF_HALO_REQUESTED): every rank receives exactly the texels its spatial-filter samples land on; the partitioned "
                                                        "frame equals the unpartitioned one bit for bit (tests/test_config5_8k.py)",
                                           "exact": "whole image (PLRF_HALO_WHOLE_IMAGE): the partitioned frame equals the unpartitioned one bit for bit",
                                           "halo": "64 trace rows per 2160 frame rows: NOT the same frame - its deviation from the unpartitioned frame grows with the frame count "
                                                   "(profiles/r05_config5_series.txt); strong_scaling_vs_1gpu_same_frame compares different images"}[args.gi_exchange])
            tile_rect[0] = rects_[rank]
        be_ = RenderBackend(w_, h_, device=local_rank)
        if band_ is not None:
            fp_ = FramePipeline(be_, w_, h_, shadow_map_res=args.shadow_res, **rect_settings(rects_[rank], w_))
            if args.python_exchange:
                tiling.Exchange(fp_, tiling.DistTransport(rank, world, device=device), h_, world, rank, rects=rects_, width=w_)  # diagnostic: torch.distributed transport driven from Python
            else:
                # the C++ host's RCCL exchange: rank 0 creates the ncclUniqueId, every rank receives it once (this broadcast is the only use of
                # torch.distributed on the data path's behalf), then ncclCommInitRank inside libplr
                uid = torch.zeros(128, dtype=torch.uint8, device=device)
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(fp_.rccl_unique_id()), dtype=torch.uint8))
                if world > 1:
                    dist.broadcast(uid, src=0)
                fp_.attach_rccl_rects(bytes(uid.cpu().numpy().tobytes()), rank, world, w_, h_, rects_)
        else:
            extra = dict(run_sky_luts=1, run_volumetrics=1, run_light_matrix=1) if args.producers else {}
            fp_ = FramePipeline(be_, w_, h_, shadow_map_res=args.shadow_res, **extra)
        # (the exact partition's denoiser weighs samples anywhere in the frame: depth and normals of the whole frame are inputs of every rank then)
        sc = build_scene(args, device, w_, h_, None if args.whole_frame_inputs else band_)
        sc[2].upload(fp_)
        be_.waitForGPUIdle()
        return be_, fp_, sc, w_, h_, band_

    tile_rect = [None]  # this rank's rectangle of the partition (make("bands"))

    # ---- strong scaling (BASELINE's ">= 3.5x at 4 GPUs on 8K" is T(the frame, 1 GPU) / T(the same frame, N GPUs)): rank 0 renders the N x larger
    # frame UNPARTITIONED first, the other ranks wait. (A frame beyond the shader's 11 pyramid levels gets the per-tile pyramid, as in tests/test_config5_8k.py.)
    single_gpu_same_frame_ms = None
    if world > 1 and not args.no_strong_scaling:
        if rank == 0:
            w1, h1 = band_frame_size()
            be1 = RenderBackend(w1, h1, device=local_rank)
            fp1 = FramePipeline(be1, w1, h1, shadow_map_res=args.shadow_res)
            _, cams1, inputs1 = build_scene(args, device, w1, h1, None)
            inputs1.upload(fp1)
            for i in range(10):
                fp1.frame(cams1[i + 1], 1.0 / 60.0, 0.5)
            be1.waitForGPUIdle()
            n1 = max(20, min(args.steps, 100))
            t1 = time.perf_counter()
            for i in range(n1):
                fp1.frame(cams1[(i % 20) + 1], 1.0 / 60.0, 0.5)
            be1.waitForGPUIdle()
            single_gpu_same_frame_ms = (time.perf_counter() - t1) * 1e3 / n1
            fp1.destroy()
            be1.shutdown()
            del inputs1
            torch.cuda.empty_cache()
        dist.barrier()

    # ---- which geometry? With --partition auto and an even N both are timed in THIS run - a short run each, same scene, same exchange mode, every
    # rank's clock, the slowest rank counts - and the faster one becomes the headline run below; the other one's time is reported under alt_partition. A geometry
    # that cannot run is reported as such and the other one is taken.
    alt_partition = None
    if world > 1 and world % 2 == 0 and args.partition == "auto":
        quick = {}
        for tiles in (False, True):
            geometry["tiles"] = tiles
            ok, ms_q, err = 1, float("inf"), None
            be_q = fp_q = None
            try:
                be_q, fp_q, (_, cams_q, _), _, _, _ = make("bands")
                for i in range(6):
                    fp_q.frame(cams_q[i + 1], 1.0 / 60.0, 0.5)
                be_q.waitForGPUIdle()
            except Exception as e:  # noqa: BLE001
                ok, err = 0, "%s: %s" % (type(e).__name__, str(e)[:200])
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                n_q = max(10, min(args.steps, 60))
                dist.barrier()
                tq = time.perf_counter()
                for i in range(n_q):
                    fp_q.frame(cams_q[(i % 20) + 1], 1.0 / 60.0, 0.5)
                be_q.waitForGPUIdle()
                tt = torch.tensor([(time.perf_counter() - tq) * 1e3 / n_q], dtype=torch.float64, device=device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ms_q = float(tt.item())
            quick[tiles] = {"kind": "tiles %dx%d" % grid() if tiles else "row bands", "ms_per_step": round(ms_q, 4) if ms_q != float("inf") else None,
                            "steps": max(10, min(args.steps, 60)), "rects": band_partition.get("rects"), "error": err if int(flag.item()) == 0 else None}
            try:
                if fp_q is not None:
                    fp_q.destroy()
                if be_q is not None:
                    be_q.shutdown()
            except Exception:  # noqa: BLE001
                pass
            torch.cuda.empty_cache()
        geometry["tiles"] = (quick[True]["ms_per_step"] or float("inf")) < (quick[False]["ms_per_step"] or float("inf"))
        alt_partition = dict(quick[not geometry["tiles"]], note="the geometry NOT taken for the headline: a short run of the same frame in this invocation")
        band_partition["chosen_from"] = {"row bands": quick[False]["ms_per_step"], "tiles": quick[True]["ms_per_step"]}

    parallelism_note = None
    if world > 1 or args.force_bands:
        # the band path needs RCCL point-to-point between neighbouring ranks; one frame is tried on every rank first. If it cannot run the
        # benchmark fails; with --allow-replicas every rank falls back to an independent 4K replica and the JSON line says so
        ok = 1
        try:
            be, fp, (scene, cams, inputs), w, h, band = make("bands")
            fp.frame(cams[1], 1.0 / 60.0, 0.5)
            be.waitForGPUIdle()
        except Exception as e:  # noqa: BLE001
            ok = 0
            parallelism_note = "%s: %s" % (type(e).__name__, str(e)[:200])
            sys.stderr.write("rank %d: band rendering unavailable (%s)\n" % (rank, parallelism_note))
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if not args.allow_replicas:
                # a scaling run that silently measured N independent frames would be measuring the wrong thing
                raise SystemExit("bench.py rank %d: the band frame could not run (%s); pass --allow-replicas to measure independent replicas instead" % (
                    rank, parallelism_note or "failed on another rank"))
            try:
                fp.destroy(); be.shutdown()
            except Exception:  # noqa: BLE001
                pass
            parallelism_note = parallelism_note or "band rendering failed on another rank"
            be, fp, (scene, cams, inputs), w, h, band = make("single")
    else:
        be, fp, (scene, cams, inputs), w, h, band = make("single")
    replicas = world > 1 and band is None
    if args.exact:
        be.setMathMode(False)

    frame_no = [0]

    def step():
        fp.frame(cams[frame_no[0] + 1], 1.0 / 60.0, 0.5 + frame_no[0] / 60.0)
        frame_no[0] += 1

    # ---- the two auxiliary measurements come FIRST, untimed: the 40 frames they render also take the GPU out of its idle clocks before the W warm-up
    # frames (measured on MI355X with --steps 20: 0.878 / 0.886 ms per frame after 5 warm-up frames from idle, 0.863 after 100, 0.861 after 400).
    # ---- host cost of one frame with the GPU idle (record + launch of the C++ pipeline through the C-ABI, nothing to wait for)
    host_idle = []
    for _ in range(20):
        be.waitForGPUIdle()
        th = time.perf_counter()
        step()
        host_idle.append((time.perf_counter() - th) * 1e3)
    be.waitForGPUIdle()
    host_idle_ms = float(np.median(host_idle))

    # ---- per-pass hipEvent timings (events recorded on the backend's launch stream) for the roofline object
    pass_ms = {}
    if args.profile_frames > 0:
        be.setPassTiming(True)
        for _ in range(args.profile_frames):
            step()
            for name, ms in be.getRenderpassTimings():
                pass_ms.setdefault(name, []).append(ms)
        be.setPassTiming(False)
    for _ in range(args.warmup):
        step()
    be.waitForGPUIdle()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_elapsed = time.perf_counter() - t0  # recording + launching only (the GPU runs behind)
    be.waitForGPUIdle()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed * 1000.0 / args.steps
    # the timed frames must be fast-set frames: an execution that fell back to the general (exact-order, port-shaped) kernel of its shader is reported
    # by the backend (plr_get_general_kernel_executions) and voids the measurement
    general_count, general_names = be.getGeneralKernelExecutions()
    if general_count and not args.exact:
        raise SystemExit("bench.py: %d execution(s) of the timed frame ran the general (exact-set) kernel instead of a fast-set kernel: %s" % (general_count, general_names))

    # per-GPU pixels: the rank's rectangle (a band renders w x rows of the frame, a tile columns x rows)
    bh = h if band is None else band[1] - band[0]
    bw = w if tile_rect[0] is None else tile_rect[0][2] - tile_rect[0][0]
    bytes_per_pass, frame_bytes = algorithmic_bytes(bw, bh, args.grid ** 2, args.sdf_res, args.shadow_res, 512, 64)
    def pass_bytes(name):
        # a fused launch (pass fusion, include/plr.h) is reported as "A + B": its compulsory traffic is the sum of its passes' - an image one pass
        # writes and the next reads back still has to be written (it is an output of the boundary). Two exceptions: "X + X" is ONE pass over two
        # row ranges (band rendering), counted once; and the fused upscale + shade at fusion level 2 neither writes nor re-reads the upscaled GI
        # images (12 B/px each way) and reads the depth buffer once instead of twice (4 B/px)
        parts = name.split(" + ")
        if len(parts) == 2 and parts[0] == parts[1]:
            parts = parts[:1]
        total = sum(bytes_per_pass.get(part, 0.0) for part in parts)
        if parts == ["Indirect lighting upscale", "Forward shading (deferred)"] and be.getPassFusion()[0] >= 2:
            total -= 28.0 * w * h
        return total
    table = []
    for name, v in pass_ms.items():
        launches = len(v) / args.profile_frames
        avg = float(np.mean(v))
        table.append((name, avg, launches, pass_bytes(name)))
    table.sort(key=lambda r: -r[1] * r[2])
    roofline = None
    if table:
        name, avg_ms, launches, nbytes = table[0]
        achieved = nbytes / (avg_ms * 1e-3) / 1e9
        default_workload = (w, h, args.grid, args.sdf_res, args.shadow_res) == (3840, 2160, 16, 64, 2048) and band is None
        traffic, traffic_src = pmc_traffic(name) if default_workload else (None, None)
        # Secondary rooflines for the resource that actually binds a gather / ALU kernel: the VALU issue floor - instructions x 2 cycles
        # per wave64 instruction on a SIMD-32 (MI355X_MICROARCH.md, Wave scheduling) / 1024 SIMDs / 2.4 GHz - and the L1 floor - one cache-line access per
        # cycle and CU (the rate the shade ablation of round 4 ran into, profiles/r04_shade_ablation.txt) - each as a fraction of the measured launch time
        counters, counters_src = pmc_counters(name) if default_workload else (None, None)
        valu_roofline = l1_roofline = None
        clock = SHADER_CLOCK_GHZ
        if counters and counters.get("SQ_INSTS_VALU"):
            # the clock the kernel actually ran at: GRBM_GUI_ACTIVE / 8 XCDs cycles of the counter pass over this run's measured launch time,
            # capped at the nominal 2.4 GHz; the floor at the 2.0 GHz of a dense VALU stream beside it
            if counters.get("GRBM_GUI_ACTIVE"):
                clock = min(counters["GRBM_GUI_ACTIVE"] / 8.0 / (avg_ms * 1e6), SHADER_CLOCK_GHZ)
            t_valu_ms = counters["SQ_INSTS_VALU"] * 2.0 / SIMDS / (clock * 1e9) * 1e3
            valu_roofline = {"bound": "valu issue", "valu_instructions_per_launch": int(counters["SQ_INSTS_VALU"]), "cycles_per_instruction_at_peak": 2, "simds": SIMDS,
                             "clock_GHz": round(clock, 3), "clock_source": "GRBM_GUI_ACTIVE / 8 XCDs (counter pass) / this run's launch time, capped at 2.4",
                             "floor_ms": round(t_valu_ms, 4), "frac": round(t_valu_ms / avg_ms, 4),
                             "frac_at_2.0GHz_dense_valu_clock": round(t_valu_ms * clock / DENSE_VALU_CLOCK_GHZ / avg_ms, 4), "source": counters_src}
        if counters and counters.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            t_l1_ms = counters["TCP_TOTAL_CACHE_ACCESSES_sum"] / CUS / (clock * 1e9) * 1e3
            l1_roofline = {"bound": "L1 (TCP) cache-line accesses", "accesses_per_launch": int(counters["TCP_TOTAL_CACHE_ACCESSES_sum"]), "accesses_per_cycle_per_cu_at_peak": 1,
                           "cus": CUS, "floor_ms": round(t_l1_ms, 4), "frac": round(t_l1_ms / avg_ms, 4), "source": counters_src}
        roofline = {"bound": "hbm", "kernel": name, "hip_kernel": (PASS_KERNEL.get(name) or PASS_KERNEL.get(name.split(" + ")[-1]) or [None])[0], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(nbytes),
                    "valu_roofline": valu_roofline, "l1_roofline": l1_roofline}
    if args.pass_table and rank == 0:
        tot = sum(r[1] * r[2] for r in table)
        sys.stderr.write("%-36s %9s %7s %9s %8s\n" % ("pass", "avg ms", "launch", "GB/s", "% frame"))
        for name, avg, launches, nbytes in table:
            sys.stderr.write("%-36s %9.4f %7.1f %9.1f %8.1f\n" % (name, avg, launches, nbytes / (avg * 1e-3) / 1e9 if avg > 0 else 0, 100 * avg * launches / tot))
        sys.stderr.write("sum of pass times %.3f ms; frame (wall) %.3f ms\n" % (tot, ms_per_step))

    # ---- the SDF trace as a line of its own: rays (= trace pixels) per second, and how many instances the culling tiles carry (sdfCameraTileCulling.comp: 32 x 32 trace
    # pixels per tile, at most 100 instances each) - the default scene has a median of one instance per tile, --scene dense tens
    trace_line = None
    if band is None:
        try:
            tile_uints = 101
            raw = be.downloadStorageBuffer(fp.storage_buffer("sdfCulledTiles"), math.ceil(w / 32) * math.ceil((h // 2) / 32) * tile_uints * 4, dtype=np.uint32).reshape(-1, tile_uints)
            tw_tiles, th_tiles = math.ceil((w // 2) / 32), math.ceil((h // 2) / 32)
            counts = raw.reshape(th_tiles, math.ceil(w / 32), tile_uints)[:, :tw_tiles, 0].reshape(-1)  # the buffer's row stride is the FULL-resolution tile count (sdfCulling.inc:17-20)
            t_ms = sum(v for k, v in ((n, float(np.mean(x))) for n, x in pass_ms.items()) if "SDF trace" in k)
            trace_line = {"rays_per_frame": (w // 2) * (h // 2), "trace_ms": round(t_ms, 4) if t_ms else None,
                          "rays_per_s": round((w // 2) * (h // 2) / (t_ms * 1e-3), 1) if t_ms else None,
                          "instances_per_culling_tile": {"median": float(np.median(counts)), "mean": round(float(counts.mean()), 2), "max": int(counts.max()),
                                                         "tiles_at_the_cap_of_100": int((counts >= 100).sum()), "tiles": int(counts.size)}}
        except Exception as e:  # noqa: BLE001
            trace_line = {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}

    exchange_stats = None
    if band is not None and not args.python_exchange:
        sent, received, groups = fp.rccl_stats()
        exchange_stats = {"rank0_bytes_sent_per_frame": sent, "rank0_bytes_received_per_frame": received, "point_to_point_groups_per_frame": groups}
        info = fp.rccl_info()
        # what the exchange ran on: communicator size and RCCL version, the ordering mode of the overlapped exchanges (2: the producers' edge
        # signal + hipStreamWaitValue32; 1: an event behind the producer, also the fallback without stream memory operations), packed rectangles or whole rows
        exchange_stats.update(rccl_ranks=info["rccl_ranks"], rccl_version=info["rccl_version"], band_overlap_exchange=info["overlap_mode"],
                              regions="rectangles through pack / unpack kernels" if info["packed_regions"] else "whole rows straight from the images",
                              stream_wait_value_supported=bool(info["stream_wait_value_supported"]), watchdog_ms=info["watchdog_ms"])
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, os.cpu_count() or 1, device)

    if rank == 0:
        out = {
            "metric": "frames/sec full GI+shade+post @4K; %HBM roofline; 1/2/4/8-GPU scaling",
            "value": round(1000.0 / ms_per_step * (w * h) / 8294400.0 * (world if replicas else 1), 3),
            "unit": "frames/s (3840x2160-equivalent: frames/s x frame pixels / 8294400)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "host_ms_per_step": round(host_elapsed * 1000.0 / args.steps, 4),
            "host_ms_per_frame_idle_gpu": round(host_idle_ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "full frame (exposure + HiZ + SDF GI trace/denoise + deferred shade + TAA + bloom + tonemap) %dx%d, %d SDF instances x %d^3, "
                                   "half-res trace, reference default settings" % (w, h, args.grid ** 2, args.sdf_res) + (" - DENSE scene (instances 1 m apart, tens of instances per culling tile): not the headline workload" if args.scene == "dense" else ""),
                       "scene": args.scene, "resolution": [w, h], "sdf_instances": args.grid ** 2, "sdf_resolution": args.sdf_res, "kernel_set": "exact" if args.exact else "fast",
                       # what the timed frames do besides the workload's size: camera translation per frame (static G-buffer, moving view for the reprojections),
                       # backend scheduling switches (include/plr.h): pass fusion level, asynchronous frame tail (bloom chain + tonemap beside the next frame)
                       "input_producers_as_compute": bool(args.producers), "camera_step_per_frame": [0.002, 0.0, 0.004], "pass_fusion": be.getPassFusion()[0], "async_tail": be.getAsyncTail()[0],
                       "general_kernel_executions_in_last_timed_frame": general_count,
                       "parallelism": ("one %dx%d frame in %s (one per GPU), halos exchanged over RCCL point-to-point (%s) "
                                       "+ one 512 B histogram all-reduce per frame" % (w, h, ("%d x %d screen tiles of ~%dx%d" % (grid()[0], grid()[1], w // grid()[0], h // grid()[1])) if grid()[0] > 1 else
                                                                                        ("%d row bands of ~%d rows" % (world, h // world)),
                                                                                        "torch.distributed from Python" if args.python_exchange else "ncclSend/ncclRecv from the C++ host")) if (world > 1 and not replicas) else
                                      ("replicas: one independent %dx%d frame per GPU (band rendering unavailable: %s)" % (w, h, parallelism_note) if replicas else "single GPU")},
            # the frame against its two bounds: the HBM roofline (algorithmic bytes / 8 TB/s) and the VALU issue floor of its instruction streams - the 60 % target
            # of the HBM roofline is ms_at_60pct_of_8TBs; a frame_valu_floor_ms above it says the target is out of reach of these kernels at any issue efficiency
            "frame_roofline": dict({"algorithmic_bytes": int(frame_bytes), "achieved_GBs": round(frame_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                    "frac_of_8TBs": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "ms_at_60pct_of_8TBs": round(frame_bytes / (0.6 * HBM_PEAK_GBS * 1e9) * 1e3, 4)},
                                   **((frame_valu_floor() or {}) if (w, h, args.grid, args.sdf_res, args.shadow_res) == (3840, 2160, 16, 64, 2048) and band is None else {})),
            "band_partition": band_partition if band is not None else None,
            "alt_partition": alt_partition,
            # strong scaling against the SAME frame on one GPU (rank 0, same run): the figure BASELINE's ">= 3.5x at 4 GPUs on 8K" is written in. Measured on
            # one node only when the driver's scaling run has N GPUs; null at N = 1
            "single_gpu_same_frame_ms": round(single_gpu_same_frame_ms, 4) if single_gpu_same_frame_ms else None,
            "strong_scaling_vs_1gpu_same_frame": round(single_gpu_same_frame_ms / ms_per_step, 4) if (single_gpu_same_frame_ms and not replicas) else None,
            "roofline": roofline,
            "sdf_trace": trace_line,
            "cpu_baseline": cpu,
            "exchange": exchange_stats,
            "passes_ms": {name: round(avg * launches, 4) for name, avg, launches, _ in table},
        }
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    fp.destroy()
    be.shutdown()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
