"""CPU mirror of the C++ FramePipeline built from the oracle passes: same pass order, same ping-pong indices, fed with the
global UBO / TAA weights / frustum the pipeline submitted. Used by the whole-frame parity test, smoke() and cpu_baseline."""
import ctypes as C
import math
import struct

import numpy as np

import passes
import pyoracle as orc
from util import F


class OracleFrame:
    def __init__(self, inputs, w, h, lut_res, settings):
        self.inp, self.w, self.h, self.lut_res, self.s = inputs, w, h, lut_res, settings
        self.half = bool(settings.sdf_half_res_trace)
        self.tw, self.th = (w // 2, h // 2) if self.half else (w, h)
        z32 = lambda n: np.zeros(n, np.uint32)
        self.color = [z32(w * h), z32(w * h)]
        self.post1 = z32(w * h)
        self.taa_hist = [z32(w * h), z32(w * h)]
        n = self.tw * self.th
        self.ind_y = [np.zeros(n * 4, np.uint16), np.zeros(n * 4, np.uint16)]
        self.ind_c = [np.zeros(n * 2, np.uint16), np.zeros(n * 2, np.uint16)]
        self.hist_y = [np.zeros(n * 4, np.uint16), np.zeros(n * 4, np.uint16)]
        self.hist_c = [np.zeros(n * 2, np.uint16), np.zeros(n * 2, np.uint16)]
        self.full_y, self.full_c = np.zeros(w * h * 4, np.uint16), np.zeros(w * h * 2, np.uint16)
        self.light = struct.pack("<5f", 0, 0, 0, 0, 0)
        self.rt_index = 0
        self.cpu_frame = 0
        self.brdf_lut = None
        self.swapchain = None
        self.tiles = None
        # oracle-side global texture array with the backend's indices
        vol_idx = inputs.volume_indices
        self._noise_idx = None
        # capture=True: frame() keeps the inputs and outputs of every pass of the LAST frame in self.cap (per-pass parity tests feed the
        # HIP passes exactly what the oracle pass consumed)
        self.capture = False
        self.cap = {}

    def _bindless(self, g):
        noise_idx = [int(x) for x in g.noiseTextureIndices]
        arr, n, keep = passes.orc_bindless(self.inp.volumes, self.inp.sdf_res, self.inp.noise, list(self.inp.volume_indices), noise_idx)
        self._keep = keep
        return arr, n

    def frame(self, global_bytes, weights9, frustum_bytes, influence):
        inp, w, h, s = self.inp, self.w, self.h, self.s
        L = orc.lib()
        g = orc.global_from_bytes(global_bytes)
        self.cpu_frame += 1
        prev = self.rt_index
        self.rt_index = (self.rt_index + 1) % 2
        cur = self.rt_index
        depth = inp.gb["depth"]
        if self.brdf_lut is None:
            self.brdf_lut = passes.orc_brdf_lut(self.lut_res, int(s.diffuse_brdf))
        if s.run_exposure:
            _, self.hist = passes.orc_histogram(self.color[prev], w, h, self.light)
            lb = passes.orc_pre_expose(self.hist, self.light, inp.transmission, 128, 128, global_bytes)
            self.light = lb.tobytes()
        if s.run_hiz:
            self.hiz = passes.orc_hiz(depth, w, h)
        if s.run_gi and s.indirect_lighting_tech == 0:
            if self.half:
                self.half_depth = passes.orc_depth_downscale(depth, w, h)
            fr = np.frombuffer(frustum_bytes, np.float32)
            fpts, fnrm = fr[:24].reshape(6, 4), fr[24:48].reshape(6, 4)
            self.culled, self.tiles = passes.orc_sdf_culling(inp.instance_bytes_patched, inp.bb_bytes, fpts, fnrm, influence, self.hiz[4], self.tw, self.th, global_bytes,
                                                             True, screen_w=w)
            arr, n = self._bindless(g)
            cascade = int(s.sun_shadow_cascade_count) - 1
            self.ind_y[0], self.ind_c[0] = passes.orc_sdf_trace(depth, inp.gb["normal"], w, h, self.tw, self.th, inp.sky, 200, 100, self.light,
                                                                inp.instance_bytes_patched, self.tiles, influence, inp.shadow_info, inp.shadow_maps[cascade],
                                                                inp.shadow_res, global_bytes, arr, n, strict=bool(s.sdf_strict_influence_radius_cutoff), cascade=cascade)
            if self.capture:
                self.cap["trace"] = dict(light=self.light, tiles=self.tiles.copy(), cascade=cascade, out=(self.ind_y[0].copy(), self.ind_c[0].copy()))
            if self.half:
                dsrc, dfmt, dw, dh = self.half_depth, F.R16_sFloat, self.tw, self.th
            else:
                dsrc, dfmt, dw, dh = depth, F.Depth32, w, h
            self.ind_y[1], self.ind_c[1] = passes.orc_gi_spatial(self.ind_y[0], self.ind_c[0], self.tw, self.th, dsrc, dfmt, dw, dh, inp.gb["normal"], w, h, global_bytes, 0)
            if self.capture:
                self.cap["spatial0"] = dict(inp=(self.ind_y[0].copy(), self.ind_c[0].copy()), depth=(dsrc, dfmt, dw, dh), out=(self.ind_y[1].copy(), self.ind_c[1].copy()))
                self.cap["temporal"] = dict(inp=(self.ind_y[1].copy(), self.ind_c[1].copy(), self.hist_y[0].copy(), self.hist_c[0].copy()))
            t = passes.orc_gi_temporal(self.ind_y[1], self.ind_c[1], self.hist_y[0], self.hist_c[0], self.tw, self.th, inp.gb["motion"], inp.gb["motion"], w, h, global_bytes)
            self.ind_y[0], self.ind_c[0], self.hist_y[1], self.hist_c[1] = t
            if self.capture:
                self.cap["temporal"]["out"] = tuple(a.copy() for a in t)
                self.cap["spatial1"] = dict(inp=(self.hist_y[1].copy(), self.hist_c[1].copy()), depth=(dsrc, dfmt, dw, dh))
            self.hist_y[0], self.hist_c[0] = passes.orc_gi_spatial(self.hist_y[1], self.hist_c[1], self.tw, self.th, dsrc, dfmt, dw, dh, inp.gb["normal"], w, h,
                                                                   global_bytes, 1)
            if self.capture:
                self.cap["spatial1"]["out"] = (self.hist_y[0].copy(), self.hist_c[0].copy())
            if self.half:
                self.full_y, self.full_c = passes.orc_gi_upscale(self.hist_y[0], self.hist_c[0], self.tw, self.th, depth, self.half_depth, w, h, global_bytes)
                if self.capture:
                    self.cap["upscale"] = dict(inp=(self.hist_y[0].copy(), self.hist_c[0].copy()), half_depth=self.half_depth, out=(self.full_y.copy(), self.full_c.copy()))
        if s.run_shading:
            arr, n = self._bindless(g)
            ysh, cocg = (self.full_y, self.full_c) if self.half else (self.hist_y[0], self.hist_c[0])
            self.color[cur] = passes.orc_deferred_shading(inp.gb, w, h, self.brdf_lut, self.lut_res, self.light, inp.shadow_info, inp.shadow_maps, inp.shadow_res, ysh, cocg,
                                                          inp.froxel, inp.froxel_dims, inp.vol_settings, inp.sky, global_bytes, arr, n, int(s.diffuse_brdf),
                                                          int(s.direct_multiscatter), bool(s.use_geometry_aa), int(s.indirect_lighting_tech),
                                                          int(s.sun_shadow_cascade_count))
            if self.capture:
                self.cap["shade"] = dict(light=self.light, gi=(ysh.copy(), cocg.copy()), out=self.color[cur].copy())
        src = self.color[cur]
        if s.run_taa and s.taa_enabled and getattr(s, "taa_use_separate_supersampling", 0):
            # TAA::computeTemporalSuperSampling (TAA.cpp:85-137): luminance of this frame, 2-frame blend into postProcessBuffers[0]
            m2 = self.cpu_frame % 2
            if not hasattr(self, "scene_lum"):
                self.scene_lum = [np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)]
            self.scene_lum[m2] = passes.orc_color_to_luminance(self.color[cur], w, h)
            self.post0 = passes.orc_temporal_supersampling(self.color[cur], self.color[prev], inp.gb["motion"], depth, depth, self.scene_lum[m2], self.scene_lum[(m2 + 1) % 2],
                                                           w, h, global_bytes, bool(s.taa_supersample_use_tonemapping))
            src = self.post0
        if s.run_taa and s.taa_enabled:
            m2 = self.cpu_frame % 2
            out, hist = passes.orc_taa(src, self.taa_hist[m2], inp.gb["motion"], depth, w, h, weights9, global_bytes, bool(s.taa_use_clipping),
                                       bool(s.taa_use_motion_vector_dilation), int(s.taa_history_sampling_tech), bool(s.taa_filter_use_tonemapping))
            if self.capture:
                self.cap["taa"] = dict(inp=src.copy(), history=self.taa_hist[m2].copy(), weights=np.array(weights9, np.float32), out=out.copy())
            self.taa_hist[(m2 + 1) % 2] = hist
            self.post1 = out
            src = self.post1
        if s.run_bloom and s.bloom_enabled:
            if self.capture:
                self.cap["bloom"] = dict(inp=src.copy())
            out, _, _ = passes.orc_bloom(src, w, h, float(s.bloom_strength), float(s.bloom_radius))
            if self.capture:
                self.cap["bloom"]["out"] = out.copy()
            if src is self.post1:
                self.post1 = out
            else:
                self.color[cur] = out
            src = out
        if s.run_tonemap:
            self.swapchain = passes.orc_tonemap(src, w, h, global_bytes)
            if self.capture:
                self.cap["tonemap"] = dict(inp=src.copy(), out=self.swapchain.copy())
        if self.capture:
            self.cap["global"] = bytes(global_bytes)
        return src
