/* C-ABI of the asset-pipeline SDF bake (BASELINE config 1 / SURVEY §8 a16, f2), executed on the GPU.
 *
 * Replaces  SceneSDFTextures computeSceneSDFTextures(const std::vector<MeshData>& meshes,
 *                                                    const std::vector<AxisAlignedBoundingBox>& AABBList)
 * (Plain/src/AssetPipeline/SceneSDF.h, implementation SceneSDF.cpp:97-157) and the per-mesh computeSDF (SceneSDF.cpp:296-513).
 * The reference runs one CPU job per mesh; here every voxel of every mesh is traced by the GPU (4x4x4 voxel bricks per
 * workgroup, 225 rays per voxel through the same 16^3 uniform triangle grid, same sign heuristic, same closest-triangle
 * fallback). Host buffers in, host buffers out: this is an offline tool boundary, not the frame path.
 * All functions return PLR_OK (0) or a negative code; plr_last_error() (plr.h) explains. */
#ifndef PLR_SDF_BAKE_H
#define PLR_SDF_BAKE_H
#include "plr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the fields of MeshData (Plain/src/Common/MeshData.h:13-24) the bake reads */
typedef struct plr_mesh_data {
    const float* positions;  /* vertex_count x 3 */
    uint32_t vertex_count;
    const uint32_t* indices; /* triangle list; triangle normal = normalize(cross(v0 - v2, v0 - v1)), SceneSDF.cpp:273 */
    uint32_t index_count;
} plr_mesh_data;

/* AxisAlignedBoundingBox, Plain/src/Common/AABB.h */
typedef struct plr_aabb { float min[3]; float max[3]; } plr_aabb;

/* the ImageDescription computeSceneSDFTextures fills per mesh (SceneSDF.cpp:120-141): per axis
 * clamp(nextPowerOfTwo(extent / 0.25), 16, 64), Type3D, R16_sFloat, Storage | Sampled, one mip */
int plr_sdf_texture_description(const plr_aabb* mesh_bounds, plr_image_desc* out_desc);

/* padSDFBoundingBox, Plain/src/Common/sdfUtilities.cpp:5-19: the box the volume actually covers */
int plr_sdf_padded_bounds(const plr_aabb* mesh_bounds, plr_aabb* out_padded);

/* computeSDF, SceneSDF.cpp:296-513, for one mesh at an explicit resolution. out_data receives width*height*depth half floats
 * (x fastest, then y, then z: flattenGridIndex, SceneSDF.cpp:233-235); out_size is its capacity in bytes. */
int plr_compute_sdf(int device, const plr_mesh_data* mesh, const plr_aabb* mesh_bounds, uint32_t width, uint32_t height,
                    uint32_t depth, void* out_data, size_t out_size);

/* computeSceneSDFTextures, SceneSDF.cpp:97-157: descriptions[i] is filled by the resolution rule, out_data[i] (capacity
 * out_sizes[i] bytes, at least 64*64*64*2 is always enough) receives the volume of mesh i. A null out_data[i] skips mesh i
 * (the reference skips meshes without an sdf texture path, SceneSDF.cpp:113-115). out_seconds (optional) = wall time. */
int plr_compute_scene_sdf_textures(int device, const plr_mesh_data* meshes, const plr_aabb* bounds, uint32_t mesh_count,
                                   plr_image_desc* out_descriptions, void* const* out_data, const size_t* out_sizes,
                                   double* out_seconds);

/* device time of the bake kernel of the last plr_compute_sdf call on this thread, in ms (benchmarks) */
int plr_sdf_last_kernel_ms(float* out_ms);

#ifdef __cplusplus
}
#endif
#endif
