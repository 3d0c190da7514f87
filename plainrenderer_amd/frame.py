"""ctypes binding of the C++ frame pipeline (include/plr_frame.h) and the upload of one synthetic scene into it.

The pipeline (plainrenderer_amd/csrc/frontend/frame_pipeline.cpp) is the host-side mirror of the reference's
RenderFrontend::prepareRenderpasses + technique classes; this module only drives it for tests and benchmarks.
"""
import ctypes as C
import struct

import numpy as np

from . import synth
from .backend import ImageHandle, PlrError, RenderBackend, _ImageHandle
from .scene import Camera


class PlrfSettings(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("width", "height", "shadow_map_res", "brdf_lut_res", "max_sdf_instances", "froxel_depth", "taa_enabled",
                                          "taa_use_clipping", "taa_use_motion_vector_dilation", "taa_history_sampling_tech", "taa_filter_use_tonemapping",
                                          "bloom_enabled")] + \
               [("bloom_strength", C.c_float), ("bloom_radius", C.c_float), ("sdf_half_res_trace", C.c_uint32), ("sdf_strict_influence_radius_cutoff", C.c_uint32),
                ("sdf_trace_influence_radius", C.c_float)] + \
               [(n, C.c_uint32) for n in ("diffuse_brdf", "direct_multiscatter", "indirect_lighting_tech", "use_geometry_aa", "sun_shadow_cascade_count",
                                          "run_exposure", "run_hiz", "run_gi", "run_shading", "run_taa", "run_bloom", "run_tonemap",
                                          "band_row_begin", "band_row_end", "band_gi_halo", "band_gi_history_halo", "band_color_halo", "band_post_halo",
                                          "run_light_matrix")] + [("volumetrics_max_distance", C.c_float), ("taa_use_separate_supersampling", C.c_uint32), ("taa_supersample_use_tonemapping", C.c_uint32),
                                                                              ("sdf_debug_mode", C.c_uint32), ("sdf_debug_tile_usage_with_hiz", C.c_uint32),
                                                                              ("sdf_debug_use_influence_radius", C.c_uint32), ("band_taa_history_halo", C.c_uint32), ("run_volumetrics", C.c_uint32),
                                                                              ("run_sky_luts", C.c_uint32), ("band_overlap_exchange", C.c_uint32), ("band_col_begin", C.c_uint32),
                                                                              ("band_col_end", C.c_uint32)]


class PlrfExchangeItem(C.Structure):
    _fields_ = [("image", _ImageHandle), ("device_ptr", C.c_void_p), ("row_begin", C.c_uint32), ("row_end", C.c_uint32), ("halo_rows", C.c_uint32),
                ("row_bytes", C.c_uint32), ("image_rows", C.c_uint32), ("col_begin", C.c_uint32), ("col_end", C.c_uint32), ("image_cols", C.c_uint32),
                ("texel_bytes", C.c_uint32)]


EXCHANGE_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)
EXCHANGE_HISTOGRAM, EXCHANGE_GI_TRACE, EXCHANGE_GI_TEMPORAL, EXCHANGE_GI_HISTORY, EXCHANGE_POST, EXCHANGE_DEPTH_APEX = range(6)
EXCHANGE_BEGIN, EXCHANGE_END, EXCHANGE_ID_MASK = 0x100, 0x200, 0xff  # phase bits (band_overlap_exchange)


class PlrfCamera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("forward", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3)]


class LocalExchangeGroup:
    """shared state of the in-process transport of the native exchange (include/plr_frame.h plrf_local_group_*): one per partition, created before its ranks'
    pipelines attach (FramePipeline.attach_local_rects), destroyed after they are gone"""

    def __init__(self, world):
        from .backend import _load
        self.lib = _load()
        self.handle = C.c_void_p()
        self.lib.plrf_local_group_destroy.argtypes = [C.c_void_p]
        self.lib.plrf_local_group_abort.argtypes = [C.c_void_p]
        self.lib.plrf_local_group_freeze.argtypes = [C.c_void_p, C.c_int]
        if self.lib.plrf_local_group_create(C.c_int(world), C.byref(self.handle)) != 0:
            raise PlrError("plrf_local_group_create failed")

    def abort(self):
        """wake ranks that wait for a peer which has failed"""
        if self.handle:
            self.lib.plrf_local_group_abort(self.handle)

    def freeze(self, frozen=True):
        """frozen: a rank copies from what its peers posted last instead of waiting for them (one partition timed alone with real neighbour data)"""
        self.lib.plrf_local_group_freeze(self.handle, C.c_int(int(bool(frozen))))

    def destroy(self):
        if self.handle:
            self.lib.plrf_local_group_destroy(self.handle)
            self.handle = None


class FramePipeline:
    def __init__(self, be: RenderBackend, width, height, **overrides):
        self.be, self.lib = be, be.lib
        self.lib.plrf_last_error.restype = C.c_char_p
        s = PlrfSettings()
        self._check(self.lib.plrf_default_settings(C.byref(s), C.c_uint32(width), C.c_uint32(height)))
        for k, v in overrides.items():
            if not hasattr(s, k):
                raise KeyError(k)
            setattr(s, k, v)
        self.settings = s
        self.handle = C.c_void_p()
        self._check(self.lib.plrf_create(C.byref(s), C.byref(self.handle)))
        self.width, self.height = width, height

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.plrf_last_error().decode() or self.lib.plr_last_error().decode()
            raise PlrError("plrf error %d: %s" % (rc, msg))

    def destroy(self):
        if self.handle:
            self.detach_rccl()
            self.lib.plrf_destroy(self.handle)
            self.handle = None

    # ---- native RCCL halo exchange (include/plr_frame.h, csrc/frontend/band_exchange.cpp): no Python in the frame loop
    def rccl_unique_id(self):
        """bytes of a fresh ncclUniqueId (rank 0 calls this and hands the bytes to every rank)"""
        buf = C.create_string_buffer(128)
        if self.lib.plrf_rccl_get_unique_id(buf) != 0:
            self.lib.plrf_rccl_last_error.restype = C.c_char_p
            raise PlrError("plrf_rccl_get_unique_id: " + self.lib.plrf_rccl_last_error().decode())
        return buf.raw

    def attach_rccl(self, unique_id, rank, world, frame_height, bounds=None):
        """bounds: world + 1 row boundaries of a partition chosen by the caller (tiling.balanced_bounds), or None for the equal partition"""
        self.lib.plrf_rccl_last_error.restype = C.c_char_p
        x = C.c_void_p()
        rows = (C.c_uint32 * (world + 1))(*[int(v) for v in bounds]) if bounds is not None else None
        rc = self.lib.plrf_rccl_attach_rows(self.handle, C.create_string_buffer(bytes(unique_id), 128), C.c_int(rank), C.c_int(world), C.c_uint32(frame_height), rows, C.byref(x))
        if rc != 0:
            raise PlrError("plrf_rccl_attach failed (%d): %s" % (rc, self.lib.plrf_rccl_last_error().decode()))
        self._rccl = x

    def attach_rccl_rects(self, unique_id, rank, world, frame_width, frame_height, rects):
        """the native exchange for a partition into rectangles [(x0, y0, x1, y1)] (tile rendering; plrf_rccl_attach_rects). unique_id None: LOOPBACK - no
        communicator, the pack / unpack kernels run and a device copy stands in for the links (single-GPU replay of one partition's frame)"""
        self.lib.plrf_rccl_last_error.restype = C.c_char_p
        x = C.c_void_p()
        flat = (C.c_uint32 * (4 * world))(*[int(v) for r in rects for v in r])
        uid = C.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
        rc = self.lib.plrf_rccl_attach_rects(self.handle, uid, C.c_int(rank), C.c_int(world), C.c_uint32(frame_width), C.c_uint32(frame_height), flat, C.byref(x))
        if rc != 0:
            raise PlrError("plrf_rccl_attach_rects failed (%d): %s" % (rc, self.lib.plrf_rccl_last_error().decode()))
        self._rccl = x

    def attach_local_rects(self, group, rank, world, frame_width, frame_height, rects):
        """the native exchange over the IN-PROCESS transport (plrf_local_attach_rects): every rank of the partition lives in this process on one GPU (a thread each);
        group: a LocalExchangeGroup shared by the ranks"""
        self.lib.plrf_rccl_last_error.restype = C.c_char_p
        x = C.c_void_p()
        flat = (C.c_uint32 * (4 * world))(*[int(v) for r in rects for v in r])
        rc = self.lib.plrf_local_attach_rects(self.handle, group.handle, C.c_int(rank), C.c_int(world), C.c_uint32(frame_width), C.c_uint32(frame_height), flat, C.byref(x))
        if rc != 0:
            raise PlrError("plrf_local_attach_rects failed (%d): %s" % (rc, self.lib.plrf_rccl_last_error().decode()))
        self._rccl = x

    def rccl_info(self):
        """dict: rccl_ranks, rccl_version, overlap_mode, packed_regions, stream_wait_value_supported, watchdog_ms (plrf_rccl_info)"""
        names = ("rccl_ranks", "rccl_version", "overlap_mode", "packed_regions", "stream_wait_value_supported", "watchdog_ms")
        info = (C.c_int32 * len(names))()
        self._check(self.lib.plrf_rccl_get_info(self._rccl, info))
        return dict(zip(names, [int(v) for v in info]))

    def rccl_self_test_rect(self, device_ptr, pitch_bytes, texel_bytes, rect, dst_xy):
        stream = C.c_void_p()
        self.be._check(self.lib.plr_get_stream(C.byref(stream)))
        x0, y0, x1, y1 = rect
        rc = self.lib.plrf_rccl_self_test_rect(self._rccl, C.c_void_p(device_ptr), C.c_uint32(pitch_bytes), C.c_uint32(texel_bytes), C.c_uint32(x0), C.c_uint32(y0), C.c_uint32(x1),
                                               C.c_uint32(y1), C.c_uint32(dst_xy[0]), C.c_uint32(dst_xy[1]), stream)
        if rc != 0:
            self.lib.plrf_rccl_last_error.restype = C.c_char_p
            raise PlrError("plrf_rccl_self_test_rect failed (%d): %s" % (rc, self.lib.plrf_rccl_last_error().decode()))

    def detach_rccl(self):
        if getattr(self, "_rccl", None):
            self.lib.plrf_rccl_detach(self.handle, self._rccl)
            self._rccl = None

    def rccl_stats(self):
        """(bytes sent, bytes received, point-to-point exchange groups) of the last frame on this rank"""
        a, b, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.plrf_rccl_get_stats(self._rccl, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def rccl_self_test(self, device_ptr, row_bytes, src_row, dst_row, rows):
        stream = C.c_void_p()
        self.be._check(self.lib.plr_get_stream(C.byref(stream)))
        rc = self.lib.plrf_rccl_self_test(self._rccl, C.c_void_p(device_ptr), C.c_uint32(row_bytes), C.c_uint32(src_row), C.c_uint32(dst_row), C.c_uint32(rows), stream)
        if rc != 0:
            raise PlrError("plrf_rccl_self_test failed (%d): %s" % (rc, self.lib.plrf_rccl_last_error().decode()))

    def image(self, name):
        h = _ImageHandle()
        self._check(self.lib.plrf_get_image(self.handle, name.encode(), C.byref(h)))
        return ImageHandle(h.type, h.index)

    def storage_buffer(self, name):
        h = C.c_uint32()
        self._check(self.lib.plrf_get_storage_buffer(self.handle, name.encode(), C.byref(h)))
        return h.value

    def uniform_buffer(self, name):
        h = C.c_uint32()
        self._check(self.lib.plrf_get_uniform_buffer(self.handle, name.encode(), C.byref(h)))
        return h.value

    def add_sdf_volume(self, res, half_data):
        a = np.ascontiguousarray(half_data)
        out = C.c_uint32()
        self._check(self.lib.plrf_add_sdf_volume(self.handle, C.c_uint32(res), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), C.byref(out)))
        return out.value

    def add_sdf_volume_dds(self, path):
        """-> (global texture array index, (width, height, depth)) of a baked SDF volume stored as DDS"""
        out = C.c_uint32()
        size = (C.c_uint32 * 3)()
        self._check(self.lib.plrf_add_sdf_volume_dds(self.handle, str(path).encode(), C.byref(out), size))
        return out.value, tuple(int(v) for v in size)

    def set_sdf_scene(self, instance_bytes, bb_bytes):
        self._check(self.lib.plrf_set_sdf_scene(self.handle, instance_bytes, C.c_size_t(len(instance_bytes)), bb_bytes, C.c_size_t(len(bb_bytes))))

    def set_sun_direction(self, d):
        self._check(self.lib.plrf_set_sun_direction(self.handle, (C.c_float * 3)(*[float(x) for x in d])))

    def set_camera_intrinsic(self, fov, near, far):
        self._check(self.lib.plrf_set_camera_intrinsic(self.handle, C.c_float(fov), C.c_float(near), C.c_float(far)))

    def set_camera_cut(self):
        self._check(self.lib.plrf_set_camera_cut(self.handle))

    def frame(self, cam: Camera, delta_time=1.0 / 60.0, time=0.0):
        c = PlrfCamera()
        for name in ("position", "forward", "up", "right"):
            setattr(c, name, (C.c_float * 3)(*[float(x) for x in getattr(cam, name)]))
        self._check(self.lib.plrf_frame(self.handle, C.byref(c), C.c_float(delta_time), C.c_float(time)))

    # ---- band rendering (include/plr_frame.h)
    def set_exchange_callback(self, fn):
        """fn(exchange_id, hip_stream_ptr) -> None; called in pass order from inside frame() where neighbouring bands' rows are needed."""
        def tramp(_user, exchange_id, stream):
            try:
                fn(int(exchange_id), stream)
                return 0
            except Exception:  # an exception cannot cross the C frames; report and fail the frame
                import traceback
                traceback.print_exc()
                return -7
        self._exchange_cb = EXCHANGE_CALLBACK(tramp)  # keep alive
        self._check(self.lib.plrf_set_exchange_callback(self.handle, self._exchange_cb, None))

    def exchange_items(self, exchange_id):
        n = C.c_uint32(16)
        items = (PlrfExchangeItem * 16)()
        self._check(self.lib.plrf_get_exchange_items(self.handle, C.c_int(exchange_id), items, C.byref(n)))
        return [items[i] for i in range(n.value)]

    def histogram_exchange(self):
        ptr, size = C.c_void_p(), C.c_size_t()
        self._check(self.lib.plrf_get_histogram_exchange(self.handle, C.byref(ptr), C.byref(size)))
        return ptr.value, size.value

    def depth_apex_exchange(self):
        """(device pointer, 8): the band's {min, max} depth range, to be all-reduced in place (EXCHANGE_DEPTH_APEX)"""
        ptr, size = C.c_void_p(), C.c_size_t()
        self._check(self.lib.plrf_get_depth_apex_exchange(self.handle, C.byref(ptr), C.byref(size)))
        return ptr.value, size.value

    def submitted_globals(self):
        buf = C.create_string_buffer(340)
        self._check(self.lib.plrf_get_submitted_globals(self.handle, buf))
        return buf.raw

    def resolve_weights(self):
        w = np.zeros(9, np.float32)
        self._check(self.lib.plrf_get_resolve_weights(self.handle, w.ctypes.data_as(C.c_void_p)))
        return w

    def cpu_frame_index(self):
        v = C.c_uint64()
        self._check(self.lib.plrf_get_cpu_frame_index(self.handle, C.byref(v)))
        return v.value


class SyntheticInputs:
    """everything the rasterised / out-of-scope passes would have produced, generated once per scene + camera"""

    def __init__(self, scene: synth.SynthScene, cam: Camera, cam_prev: Camera, width, height, sdf_res, shadow_res, froxel_depth, sun_direction, cascades=3,
                 rows=None, depth_range=None):
        """rows=(r0, r1): generate the per-pixel inputs for these rows only (band rendering); depth_range=(min, max) linear depth of
        the WHOLE frame then has to be given so that every band fits the same shadow cascades"""
        self.width, self.height, self.sdf_res, self.shadow_res = width, height, sdf_res, shadow_res
        self.rows = rows
        self.gb = scene.gbuffer(cam, width, height, cam_prev, rows=rows)
        self.instance_bytes, self.bb_bytes, self.volumes = scene.sdf_instances(sdf_res)
        self.noise = synth.blue_noise_standins()
        self.sky = synth.sky_lut()
        self.transmission = synth.transmission_lut()
        sun = np.asarray(sun_direction, np.float64)
        self.sun = sun / np.linalg.norm(sun)
        depth = self.gb["depth"]
        n, f = cam.near, cam.far
        vis = depth[depth > 0]
        lin = n * f / (f + (1.0 - vis.astype(np.float64)) * (n - f)) if vis.size else np.array([1.0, 50.0])
        self.depth_range = depth_range if depth_range is not None else (float(lin.min()), float(lin.max()))
        self.shadow_info, self.shadow_maps = scene.shadow_cascades(cam, self.sun, self.depth_range[0], self.depth_range[1], shadow_res, cascade_count=cascades)
        self.froxel, self.froxel_dims = synth.froxel_volume(width, height, froxel_depth)
        self.vol_settings = synth.volumetric_settings_bytes(30.0)

    # ---- fixtures: the generated arrays as a flat dict of numpy arrays and back (tests/golden)
    def to_arrays(self):
        d = {"dims": np.array([self.width, self.height, self.sdf_res, self.shadow_res], np.int64), "sun": np.asarray(self.sun, np.float64),
             "instance_bytes": np.frombuffer(self.instance_bytes, np.uint8), "bb_bytes": np.frombuffer(self.bb_bytes, np.uint8),
             "shadow_info": np.frombuffer(bytes(self.shadow_info), np.uint8), "vol_settings": np.frombuffer(bytes(self.vol_settings), np.uint8),
             "sky": self.sky, "transmission": self.transmission, "froxel": self.froxel, "froxel_dims": np.asarray(self.froxel_dims, np.int64),
             "n_volumes": np.array([len(self.volumes)], np.int64)}
        for k, v in self.gb.items():
            d["gb_" + k] = v
        for i, v in enumerate(self.volumes):
            d["volume_%d" % i] = v
        for i in range(4):
            d["shadow_map_%d" % i] = self.shadow_maps[i]
            d["noise_%d" % i] = self.noise[i]
        return d

    @classmethod
    def from_arrays(cls, d):
        self = cls.__new__(cls)
        self.width, self.height, self.sdf_res, self.shadow_res = (int(v) for v in d["dims"])
        self.sun = np.asarray(d["sun"], np.float64)
        self.instance_bytes, self.bb_bytes = d["instance_bytes"].tobytes(), d["bb_bytes"].tobytes()
        self.shadow_info, self.vol_settings = d["shadow_info"].tobytes(), d["vol_settings"].tobytes()
        self.sky, self.transmission, self.froxel = d["sky"], d["transmission"], d["froxel"]
        self.froxel_dims = tuple(int(v) for v in d["froxel_dims"])
        self.gb = {k[3:]: d[k] for k in d if k.startswith("gb_")}
        self.volumes = [d["volume_%d" % i] for i in range(int(d["n_volumes"][0]))]
        self.shadow_maps = [d["shadow_map_%d" % i] for i in range(4)]
        self.noise = [d["noise_%d" % i] for i in range(4)]
        return self

    def upload(self, fp: FramePipeline):
        be = fp.be
        gb = self.gb
        r0 = self.rows[0] if getattr(self, "rows", None) is not None else None
        up = (lambda img, a: be.uploadImageRows(img, r0, a)) if r0 is not None else be.uploadImage
        for i in (0, 1):
            up(fp.image("depth%d" % i), gb["depth"])
            up(fp.image("motion%d" % i), gb["motion"])
        up(fp.image("normal"), gb["normal"])
        up(fp.image("albedo"), gb["albedo"])
        up(fp.image("specular"), gb["specular"])
        be.uploadImage(fp.image("skyLut"), self.sky)
        be.uploadImage(fp.image("transmissionLut"), self.transmission)
        be.uploadImage(fp.image("volumetricIntegrationVolume"), self.froxel)
        for i in range(4):
            be.uploadImage(fp.image("shadow%d" % i), self.shadow_maps[i])
            be.uploadImage(fp.image("noise%d" % i), self.noise[i])
        be.setStorageBufferData(fp.storage_buffer("sunShadowInfo"), self.shadow_info)
        be.setUniformBufferData(fp.uniform_buffer("volumetricSettings"), self.vol_settings)
        self.volume_indices = [fp.add_sdf_volume(self.sdf_res, v) for v in self.volumes]
        inst = bytearray(self.instance_bytes)
        for i, ti in enumerate(self.volume_indices):
            struct.pack_into("<I", inst, 16 + i * 96 + 12, ti)
        self.instance_bytes_patched = bytes(inst)
        fp.set_sdf_scene(self.instance_bytes_patched, self.bb_bytes)
        fp.set_sun_direction(self.sun)
