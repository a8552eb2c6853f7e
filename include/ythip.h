/*
 * ythip.h — C ABI of libythip.so, the MI355X (gfx950) path-tracing core that
 * sits behind Yocto/GL's `trace_samples` boundary.
 *
 * Everything crossing this boundary is plain-old-data: pointers, sizes and
 * fixed-width scalars.  No STL, no torch types.  Error convention: every entry
 * point returns 0 on success and a non-zero YTHIP_ERR_* code otherwise;
 * `ythip_last_error()` returns a human-readable message.  The C++ shim
 * (yocto-gl_amd/host/yocto_hiptrace.cpp) turns non-zero codes into
 * std::runtime_error, mirroring the reference's exception behaviour
 * (libs/yocto/yocto_trace.cpp:1437,1454,1682-1690).
 *
 * Threads: a ythip_ctx (and a ythip_multi) is used by ONE host thread at a time
 * — its stream, its pinned transfer ring and its error string are not locked; the
 * C++ shim serialises its calls under a mutex, as the reference's trace_start worker
 * serialises its own (yocto_trace.cpp:1627-1656).  The one exception is
 * ythip_cancel, which may be called from any thread while a batch runs.  Different
 * contexts are independent.  The file readers at the end of this header keep no
 * global state (ythip_io_last_error is per thread).
 *
 * Each declaration cites the reference interface (file:line under
 * /root/reference) that it replaces or mirrors.
 */
#ifndef YTHIP_H
#define YTHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YTHIP_OK 0
#define YTHIP_ERR_INVALID 1  /* bad argument (std::invalid_argument in the reference) */
#define YTHIP_ERR_HIP 2      /* a HIP runtime call failed                             */
#define YTHIP_ERR_STATE 3    /* call order violated (e.g. trace before upload)        */
#define YTHIP_ERR_SAMPLER 4  /* "sampler unknown" (yocto_trace.cpp:1437)              */
#define YTHIP_ERR_CANCELLED 5 /* stop flag observed (yocto_trace.cpp:1637)            */

#define YTHIP_INVALIDID (-1) /* yocto_math.h `invalidid` */

/* ------------------------------------------------------------------------- */
/* Flat scene: POD mirrors of the reference's data model                      */
/* ------------------------------------------------------------------------- */

/* frame3f, column-major {x, y, z, o} — libs/yocto/yocto_math.h:770-777 */
typedef struct ythip_frame {
  float x[3], y[3], z[3], o[3];
} ythip_frame;

/* camera_data — libs/yocto/yocto_scene.h:83-91 (72 B, bool widened to int32) */
typedef struct ythip_camera {
  ythip_frame frame;
  int32_t     orthographic;
  float       lens, film, aspect, focus, aperture;
} ythip_camera;

/* instance_data — libs/yocto/yocto_scene.h:144-149 (56 B) */
typedef struct ythip_instance {
  ythip_frame frame;
  int32_t     shape, material;
} ythip_instance;

/* material_type — libs/yocto/yocto_scene.h:106-111 */
enum {
  YTHIP_MATTE = 0, YTHIP_GLOSSY, YTHIP_REFLECTIVE, YTHIP_TRANSPARENT,
  YTHIP_REFRACTIVE, YTHIP_SUBSURFACE, YTHIP_VOLUMETRIC, YTHIP_GLTFPBR
};

/* material_data — libs/yocto/yocto_scene.h:122-141 (84 B, same field order) */
typedef struct ythip_material {
  int32_t type;
  float   emission[3];
  float   color[3];
  float   roughness, metallic, ior;
  float   scattering[3];
  float   scanisotropy, trdepth, opacity;
  int32_t emission_tex, color_tex, roughness_tex, scattering_tex, normal_tex;
} ythip_material;

/* environment_data — libs/yocto/yocto_scene.h:152-157 (64 B) */
typedef struct ythip_environment {
  ythip_frame frame;
  float       emission[3];
  int32_t     emission_tex;
} ythip_environment;

/* texture_data — libs/yocto/yocto_scene.h:95-103.  Texels live in one of the
 * two concatenated pools of ythip_scene (`pixelsf` vec4f / `pixelsb` vec4b);
 * `offset` is in texels. */
typedef struct ythip_texture {
  int32_t width, height, linear, nearest, clamp, is_float;
  int64_t offset;
} ythip_texture;

/* shape_data — libs/yocto/yocto_shape.h:74-88.  All shapes share concatenated
 * element / vertex pools; offsets are in elements (resp. vertices) and are -1
 * for an absent attribute.  Element indices stay shape-local, as in the
 * reference. */
typedef struct ythip_shape {
  int64_t points_offset, lines_offset, triangles_offset, quads_offset;
  int64_t positions_offset, normals_offset, texcoords_offset, colors_offset,
      radius_offset;
  int32_t num_points, num_lines, num_triangles, num_quads;
  int32_t num_positions, num_normals, num_texcoords, num_colors, num_radius;
  int32_t pad_;
} ythip_shape;

/* scene_data — libs/yocto/yocto_scene.h:191-213 (the subset trace_samples
 * reads: cameras, instances, environments, shapes, textures, materials). */
typedef struct ythip_scene {
  int32_t num_cameras, num_instances, num_environments, num_shapes,
      num_textures, num_materials;
  const ythip_camera*      cameras;
  const ythip_instance*    instances;
  const ythip_environment* environments;
  const ythip_shape*       shapes;
  const ythip_texture*     textures;
  const ythip_material*    materials;
  /* element pools */
  int64_t        num_points, num_lines, num_triangles, num_quads;
  const int32_t* points;    /* 1 int / point    */
  const int32_t* lines;     /* 2 int / line     */
  const int32_t* triangles; /* 3 int / triangle */
  const int32_t* quads;     /* 4 int / quad     */
  /* vertex pools */
  int64_t      num_positions, num_normals, num_texcoords, num_colors,
      num_radius;
  const float* positions; /* 3 float */
  const float* normals;   /* 3 float */
  const float* texcoords; /* 2 float */
  const float* colors;    /* 4 float */
  const float* radius;    /* 1 float */
  /* texel pools */
  int64_t        num_pixelsf, num_pixelsb;
  const float*   pixelsf; /* 4 float / texel */
  const uint8_t* pixelsb; /* 4 byte  / texel */
} ythip_scene;

/* bvh_node — libs/yocto/yocto_shape.h:474-480 (32 B, identical layout) */
typedef struct ythip_bvh_node {
  float   bbox_min[3], bbox_max[3];
  int32_t start;
  int16_t num;
  int8_t  axis;
  uint8_t internal;
} ythip_bvh_node;

/* scene_bvh — libs/yocto/yocto_bvh.h:70-79.  `num_trees` = num_shapes + 1;
 * tree t < num_shapes is shape t's bvh_tree, tree num_shapes is the instance
 * tree.  node_offset/prim_offset have num_trees + 1 entries. */
typedef struct ythip_bvh {
  int32_t               num_trees;
  const int64_t*        node_offset;
  const int64_t*        prim_offset;
  const ythip_bvh_node* nodes;
  const int32_t*        primitives;
} ythip_bvh;

/* trace_light / trace_lights — libs/yocto/yocto_trace.h:126-135 */
typedef struct ythip_light {
  int32_t instance, environment;
  int64_t cdf_offset;
  int32_t cdf_count, pad_;
} ythip_light;
typedef struct ythip_lights {
  int32_t            num_lights;
  const ythip_light* lights;
  int64_t            num_cdf;
  const float*       cdf;
} ythip_lights;

/* trace_sampler_type / trace_falsecolor_type — yocto_trace.h:71-90 */
enum {
  YTHIP_SAMPLER_PATH = 0, YTHIP_SAMPLER_PATHDIRECT, YTHIP_SAMPLER_PATHMIS,
  YTHIP_SAMPLER_PATHTEST, YTHIP_SAMPLER_NAIVE, YTHIP_SAMPLER_EYELIGHT,
  YTHIP_SAMPLER_DIAGRAM, YTHIP_SAMPLER_FURNACE, YTHIP_SAMPLER_FALSECOLOR
};
enum {
  YTHIP_FC_POSITION = 0, YTHIP_FC_NORMAL, YTHIP_FC_FRONTFACING,
  YTHIP_FC_GNORMAL, YTHIP_FC_GFRONTFACING, YTHIP_FC_TEXCOORD, YTHIP_FC_MTYPE,
  YTHIP_FC_COLOR, YTHIP_FC_EMISSION, YTHIP_FC_ROUGHNESS, YTHIP_FC_OPACITY,
  YTHIP_FC_METALLIC, YTHIP_FC_DELTA, YTHIP_FC_INSTANCE, YTHIP_FC_SHAPE,
  YTHIP_FC_MATERIAL, YTHIP_FC_ELEMENT, YTHIP_FC_HIGHLIGHT
};

/* trace_params — libs/yocto/yocto_trace.h:95-113 (same fields, bools as int32) + one field of this library:
 *   fastmath  0 (default): every float of trace_state is the reference's, bit for bit.
 *             1: the TOLERANCE mode — the same integrators, rng streams and traversal (hit records stay
 *             bit-exact for a given ray), but shading / sampling / camera arithmetic on the GPU's fast
 *             forms (reciprocals for divisions, hardware sin / cos / exp / log / sqrt, fused multiply-adds).
 *             The image agrees with the reference's statistically (mean within 0.5 %, 8x8-block error
 *             below the reference's own seed-to-seed spread: tests/test_gpu_fastmath.py), not bit for bit.
 *             A tree too deep for the wide walk's stack, a context forced to the binary walk
 *             (ythip_set_traversal 0) and the debug samplers (diagram, falsecolor) render with the exact
 *             kernels either way; ythip_last_launch_fastmath tells which mode the last launch ran.
 *             2: the OWN-TREE mode — the tolerance mode's arithmetic AND a traversal that no longer follows
 *             the reference's trees: the SAH tree of the device builder, two levels per 64-B node of 8-bit
 *             boxes (ythip_build_own_bvh must have been called; without it trace_samples fails).  Radiance
 *             within the same stated tolerance of the reference (tests/test_gpu_own_tree.py, against
 *             oracle/_ref); hit records equal to the reference's except at exact ties and box-edge grazes.
 *             ythip_intersect_batch is not affected by this field: hit indices of a ray batch stay bit-exact.
 * The struct is 80 bytes (the reference's trace_params + fastmath): a caller built against another layout must not
 * pass it — check ythip_abi_version() == YTHIP_ABI_VERSION and ythip_params_size() == sizeof(ythip_params). */
#define YTHIP_ABI_VERSION 6
typedef struct ythip_params {
  int32_t  camera, resolution, sampler, falsecolor, samples, bounces;
  float    clamp;
  int32_t  nocaustics, envhidden, tentfilter;
  uint64_t seed;
  int32_t  embreebvh, highqualitybvh, noparallel, pratio, denoise, batch;
  int32_t  fastmath;
} ythip_params;

/* scene_intersection — libs/yocto/yocto_bvh.h:96-102 (24 B) */
typedef struct ythip_hit {
  int32_t instance, element;
  float   u, v, distance;
  int32_t hit;
} ythip_hit;

/* ray3f — libs/yocto/yocto_geometry.h:135-140 (32 B) */
typedef struct ythip_ray {
  float o[3], d[3], tmin, tmax;
} ythip_ray;

/* Measurement record filled by ythip_get_stats (no reference equivalent; the
 * reference only prints wall-clock, yocto_cli.h:128-140). */
typedef struct ythip_stats {
  /* k_trace (the persistent extend + shade kernel, one launch per
   * trace_samples call), measured with hipEvents on the launch stream while
   * profiling bit 0 is set */
  int64_t trace_launches;
  double  trace_ms;
  /* traversal work counters (valid after a run with counting enabled) */
  int64_t rays;        /* intersect_scene_bvh calls        (yocto_bvh.cpp:554) */
  int64_t nodes;       /* BVH node pops, TLAS+BLAS         (:487,:581)         */
  int64_t triangles;   /* leaf triangle tests              (:526-533)          */
  int64_t quads;       /* leaf quad tests                  (:536-544)          */
  int64_t lines;       /* leaf line tests                  (:516-524)          */
  int64_t points;      /* leaf point tests                 (:506-514)          */
  int64_t instances;   /* TLAS leaf entries                (:600-604)          */
  int64_t shades;      /* surface interactions shaded      (yocto_trace.cpp:494-496) */
  int64_t samples;     /* trace_sample calls               (yocto_trace.cpp:1461)    */
} ythip_stats;

typedef struct ythip_ctx ythip_ctx;

/* ------------------------------------------------------------------------- */
/* Context                                                                     */
/* ------------------------------------------------------------------------- */

/* One context per process/GPU.  Mirrors make_cutrace_context
 * (libs/yocto/yocto_cutrace.h:88, yocto_cutrace.cpp:385-520). */
int         ythip_abi_version(void);  /* YTHIP_ABI_VERSION of the library that was loaded */
int         ythip_params_size(void);  /* its sizeof(ythip_params) */
int         ythip_create(int device, ythip_ctx** out);
void        ythip_destroy(ythip_ctx* ctx);
const char* ythip_last_error(const ythip_ctx* ctx); /* ctx may be NULL */
/* Launch stream (a hipStream_t); NULL = the context's own stream. */
int ythip_set_stream(ythip_ctx* ctx, void* hip_stream);
int ythip_sync(ythip_ctx* ctx);

/* ------------------------------------------------------------------------- */
/* Scene / BVH / lights residency                                              */
/* ------------------------------------------------------------------------- */

/* Flatten-and-upload of scene_data; mirrors make_cutrace_scene
 * (yocto_cutrace.cpp:564-702).  Host pointers; copied. */
int ythip_upload_scene(ythip_ctx* ctx, const ythip_scene* scene);

/* Scene ingest straight into the flat layout (the loaders of
 * yocto_sceneio.cpp produce vector-of-vectors scene_data): ythip_scene_staging allocates
 * PINNED host pools sized by counts->num_* and returns them in *staged (same struct, pointers
 * now writable: cast the const away); the loader fills cameras, instances, environments,
 * materials, textures, the per-shape descriptors and the concatenated pools IN PLACE — one
 * pass over the source, no intermediate copy —, then ythip_upload_scene_staged() sends them
 * by asynchronous DMA and keeps the pools as the host copies the BVH / light builders read.
 * Equivalent to ythip_upload_scene() of the same content (tested byte for byte). */
int ythip_scene_staging(ythip_ctx* ctx, const ythip_scene* counts, ythip_scene* staged);
int ythip_upload_scene_staged(ythip_ctx* ctx);
/* Re-upload only the cameras of the resident scene (num must equal the resident
 * count).  The interactive loop edits scene.cameras[params.camera] between
 * trace_samples calls (apps/ytrace.cpp:189-204, 248-254) without touching the
 * geometry; everything else stays resident. */
int ythip_update_cameras(ythip_ctx* ctx, const ythip_camera* cameras, int num);
/* In-place edits of the other small pools between batches (the reference reads
 * scene.materials / scene.environments fresh on every trace_samples call,
 * yocto_trace.cpp:1595-1612; its GUI edits them in place): same counts as the
 * resident scene, texture references validated.  ythip_update_materials also
 * re-derives the kernel specialisation from the new material types. */
int ythip_update_materials(ythip_ctx* ctx, const ythip_material* materials, int num);
int ythip_update_environments(ythip_ctx* ctx, const ythip_environment* environments, int num);

/* make_trace_bvh → make_scene_bvh (yocto_trace.cpp:88-96,
 * yocto_bvh.cpp:238-302,321-396): host-side build that reproduces the
 * reference's node order bit-for-bit, then upload. */
int ythip_build_bvh(ythip_ctx* ctx, const ythip_scene* scene, int highquality);
/* The own tree of ythip_params::fastmath = 2 (see ythip_params): built NEXT TO the resident reference tree, which
 * stays what every other mode, ythip_intersect_batch and ythip_bvh_download use.  Anything that changes geometry or the
 * reference tree drops it (upload_scene, build / upload / update_bvh, update_shape_vertices, update_instance_frames).
 * scene may be NULL: the geometry as resident (the context's host copies, which follow the update_* calls).
 * ythip_own_bvh_info: its node count (64 B each), leaf records (16 B each) and build times; YTHIP_ERR_STATE without one. */
int ythip_build_own_bvh(ythip_ctx* ctx, const ythip_scene* scene);
/* Where make_bvh runs.  mode 1 (default): shapes with >= min_prims primitives
 * (default 16384; <= 0 keeps the current value) are built ON THE DEVICE by a
 * level-synchronous restatement of make_bvh + split_middle / split_sah
 * (`highquality`) that reproduces the reference's node order, `primitives`
 * permutation and boxes bit for bit (yocto_bvh.cpp:108-164, 202-302; SURVEY.md
 * §8(f) rank 1); small shapes and the instance tree use the host builder (same
 * trees).  mode 0: host only.
 * Either way ythip_bvh_download returns the reference's tree. */
int ythip_set_bvh_builder(ythip_ctx* ctx, int mode, int64_t min_prims);
/* What the last ythip_build_bvh did (zeroed by ythip_upload_bvh). */
typedef struct ythip_build_info {
  int32_t device_trees, host_trees, fallbacks, max_depth;
  int64_t device_prims;
  double  device_ms; /* device time inside the builder kernels (hipEvents)        */
  double  build_ms;  /* wall time of the whole tree construction (all shapes+TLAS) */
  double  bake_ms;   /* wall time of baking pairs / leaf data / instance records   */
  int32_t host_threads; /* worker threads that built the small shapes (0: the calling thread) */
  int32_t device_tlas;  /* 1: the instance tree was built on the device                      */
} ythip_build_info;
int ythip_bvh_build_info(ythip_ctx* ctx, ythip_build_info* info);
int ythip_own_bvh_info(ythip_ctx* ctx, int64_t* num_nodes, int64_t* num_leaf4, ythip_build_info* info);
/* The baked traversal arrays, for tests: pairs are 64-B records
 * (num_pairs), quads 128-B records (num_pairs), leaf data 16-B records. */
int ythip_bvh_baked_sizes(ythip_ctx* ctx, int64_t* num_pairs, int64_t* num_leaf4);
int ythip_bvh_baked_download(ythip_ctx* ctx, float* pairs, float* leafdata, float* quads);
/* update_scene_bvh / update_shape_bvh — libs/yocto/yocto_bvh.h:87-90,
 * yocto_bvh.cpp:304-319 (refit_bvh), 398-451: after an edit that moves vertices
 * or instances but keeps every element list, the resident trees keep their
 * topology and only the boxes are recomputed, bottom-up, bit for bit the
 * reference's.  Protocol (the scene edits of an interactive session):
 *   ythip_update_shape_vertices(...)    per moved shape: new positions / normals / radii
 *                                       (any of the three may be NULL = unchanged)
 *   ythip_update_instance_frames(...)   moved instances
 *   ythip_update_bvh(...)               refit the listed shapes' trees (on the
 *                                       device when the tree was built there), then the
 *                                       instance tree, then re-bake the traversal arrays
 * `updated_instances` is accepted for signature parity and — like the reference
 * (yocto_bvh.cpp:441-447 recomputes every instance box) — not used to skip work.
 * Counts must equal the resident shape's (YTHIP_ERR_INVALID otherwise).  Lights
 * are the caller's business, as in the reference (make_trace_lights again when
 * emissive geometry moved).  ythip_bvh_build_info afterwards: device_trees /
 * host_trees = trees refitted on either side, device_ms / build_ms / bake_ms of
 * the update. */
int ythip_update_shape_vertices(ythip_ctx* ctx, int32_t shape, const float* positions, int64_t num_positions,
    const float* normals, int64_t num_normals, const float* radius, int64_t num_radius);
int ythip_update_instance_frames(ythip_ctx* ctx, const int32_t* instances, int32_t num, const ythip_frame* frames);
int ythip_update_bvh(ythip_ctx* ctx, const int32_t* updated_instances, int32_t num_instances,
    const int32_t* updated_shapes, int32_t num_shapes);

/* Upload a tree built elsewhere (e.g. by the reference's make_trace_bvh). */
int ythip_upload_bvh(ythip_ctx* ctx, const ythip_bvh* bvh);
/* Read back the resident tree (for tree-identity tests). */
int ythip_bvh_sizes(ythip_ctx* ctx, int32_t* num_trees, int64_t* num_nodes,
    int64_t* num_prims);
int ythip_bvh_download(ythip_ctx* ctx, int64_t* node_offset,
    int64_t* prim_offset, ythip_bvh_node* nodes, int32_t* primitives);

/* The same builder without a GPU context (host-only; lets the `-m "not gpu"`
 * tests pin the tree against the reference's make_scene_bvh node-for-node). */
typedef struct ythip_hostbvh ythip_hostbvh;
int  ythip_host_bvh_build(const ythip_scene* scene, int highquality,
     ythip_hostbvh** out);
/* update_scene_bvh (yocto_bvh.cpp:434-451) on a host tree: `scene` is the edited
 * scene (same element lists), the listed shapes' trees and the instance tree are
 * refitted in place. */
int  ythip_host_bvh_refit(ythip_hostbvh* bvh, const ythip_scene* scene,
     const int32_t* updated_shapes, int32_t num_shapes);
int  ythip_host_bvh_view(const ythip_hostbvh* bvh, ythip_bvh* view);
void ythip_host_bvh_free(ythip_hostbvh* bvh);

/* make_trace_lights (yocto_trace.cpp:1528-1581): host-side CDF build + upload. */
int ythip_build_lights(ythip_ctx* ctx, const ythip_scene* scene);
int ythip_upload_lights(ythip_ctx* ctx, const ythip_lights* lights);
typedef struct ythip_hostlights ythip_hostlights;
int  ythip_host_lights_build(const ythip_scene* scene, ythip_hostlights** out);
int  ythip_host_lights_view(const ythip_hostlights* lights, ythip_lights* view);
void ythip_host_lights_free(ythip_hostlights* lights);
int ythip_lights_sizes(ythip_ctx* ctx, int32_t* num_lights, int64_t* num_cdf);
int ythip_lights_download(ythip_ctx* ctx, ythip_light* lights, float* cdf);

/* ------------------------------------------------------------------------- */
/* trace_state                                                                 */
/* ------------------------------------------------------------------------- */

/* Image size rule of make_trace_state (yocto_trace.cpp:1499-1505). */
int ythip_state_size(const ythip_camera* camera, int resolution, int* width,
    int* height);
/* Per-pixel PCG seeding of make_trace_state (yocto_trace.cpp:1512-1515):
 * a serial master stream; fills 2*n uint64 {state, inc}.  Host only. */
int ythip_make_rngs(uint64_t seed, int64_t n, uint64_t* rngs);

/* Device-resident mirror of trace_state (yocto_trace.h:147-157) for the rows
 * [row_begin, row_end) of a width x height image (row sharding, §8e).
 * Arrays hold width*(row_end-row_begin) pixels.  Zero-initialises
 * image/albedo/normal/hits; rngs must be uploaded. */
int ythip_state_create(ythip_ctx* ctx, int width, int height, int row_begin,
    int row_end);
/* The same for a COLUMN-STRIPED slice (multi-GPU load balance, SURVEY.md §8e
 * "interleaved bands"): of rows [row_begin, row_end) the state holds the
 * 16-pixel-wide tile columns col_first, col_first + col_stride, ... of the frame,
 * laid side by side in a row-major local image ythip_state_local_width() pixels
 * wide.  Rank r of G uses (col_first = r, col_stride = G): every rank sees the
 * whole vertical extent of the frame (sky and ground alike), which contiguous
 * row blocks do not.  ythip_state_create == (col_first 0, col_stride 1).  The
 * reference's unit of parallel work is a pixel (yocto_trace.cpp:1600-1612);
 * which pixels a worker takes is free. */
int ythip_state_create_striped(ythip_ctx* ctx, int width, int height,
    int row_begin, int row_end, int col_first, int col_stride);
/* Pixels per row of such a slice (-1 on bad arguments).  Host only. */
int ythip_state_local_width(int width, int col_first, int col_stride);
/* Any pointer may be NULL to skip that array. */
int ythip_state_upload(ythip_ctx* ctx, const float* image, const float* albedo,
    const float* normal, const int32_t* hits, const uint64_t* rngs,
    int samples);
int ythip_state_download(ythip_ctx* ctx, float* image, float* albedo,
    float* normal, int32_t* hits, uint64_t* rngs, int* samples);
/* get_image (yocto_trace.cpp:1694-1708): only trace_state.image, width*rows vec4f,
 * linear.  The download point of a viewer (16 B/pixel instead of the 60 B/pixel
 * of ythip_state_download). */
int ythip_get_image(ythip_ctx* ctx, float* image);
/* get_albedo_image / get_normal_image (yocto_trace.h:187-190,
 * yocto_trace.cpp:1769-1791): the denoiser's guide buffers as images — width*rows
 * vec4f {albedo.xyz | normal.xyz, 1}, expanded on the device from the vec3f running
 * means trace_sample keeps (the hand-off to a denoiser).
 * get_rendered_image (:1711-1721) is ythip_get_image; get_denoised_image
 * (:1724-1766) without OIDN is get_rendered_image. */
int ythip_get_albedo_image(ythip_ctx* ctx, float* image);
int ythip_get_normal_image(ythip_ctx* ctx, float* image);
/* tonemap_image (yocto_image.cpp:911-922; tonemap yocto_color.h:355-364) of the
 * resident image ON THE DEVICE: exposure in stops, optional filmic curve, optional
 * sRGB encoding.  `ldr` (vec4f per pixel) and/or `ldr_bytes` (vec4b per pixel,
 * float_to_byte) receive the result; either may be NULL.  A display loop
 * (apps/ytrace.cpp:206-216) then moves 4 B/pixel per refresh. */
int ythip_tonemap_image(ythip_ctx* ctx, float exposure, int filmic, int srgb,
    float* ldr, uint8_t* ldr_bytes);

/* denoise_image (yocto_trace.cpp:1794-1872) and the hand-off at the end of trace_samples
 * (:1615-1618) — SURVEY.md §8(f) rank 3.  The reference's filter is Intel OIDN (a neural
 * network, not vendored) or, in the default build, a copy; its interface — HDR colour plus the
 * first-hit albedo / normal means — is what this keeps.  The filter is an edge-avoiding à-trous
 * wavelet on the albedo-demodulated colour driven by those guides (csrc/yt_denoise.h has the
 * complete definition; tests restate it in numpy).  NOT a parity feature: there is no
 * reference output to match.  `params` NULL = ythip_denoise_default_params.
 *   ythip_denoise_image: host buffers in the reference's layouts (render vec4f, albedo vec3f,
 *     normal vec3f, denoised vec4f; width*height each), any context.
 *   ythip_denoise_state: on the resident whole-frame trace_state (a sliced state is refused:
 *     gather first); the result stays on the device (ythip_state_device_denoised) until the
 *     state changes and is also copied to `denoised` when that is not NULL. */
typedef struct ythip_denoise_params {
  int32_t levels;        /* à-trous levels, tap spacing 1, 2, 4, ...  (default 5) */
  float   sigma_color;   /* tolerance on the relative colour distance, halved per level (4) */
  float   sigma_normal;  /* tolerance on |n_p - n_q|  (0.35) */
  float   sigma_albedo;  /* tolerance on |albedo_p - albedo_q|  (0.1) */
} ythip_denoise_params;
void ythip_denoise_default_params(ythip_denoise_params* params);
int  ythip_denoise_image(ythip_ctx* ctx, const ythip_denoise_params* params, int32_t width, int32_t height,
     const float* render, const float* albedo, const float* normal, float* denoised);
int  ythip_denoise_state(ythip_ctx* ctx, const ythip_denoise_params* params, float* denoised);
int  ythip_state_device_denoised(ythip_ctx* ctx, void** image);
/* Use caller-owned DEVICE buffers (e.g. torch tensors) for the state arrays,
 * so that the framebuffer gather can run on them directly (RCCL). */
int ythip_state_bind_device(ythip_ctx* ctx, void* image, void* albedo,
    void* normal, void* hits, void* rngs);
int ythip_state_set_samples(ythip_ctx* ctx, int samples);
int ythip_state_get_samples(ythip_ctx* ctx, int* samples);
/* The device address of the resident `image` array (vec4f per pixel of the slice), after
 * draining the context's stream: what a framebuffer gather (RCCL) reads. */
int ythip_state_device_image(ythip_ctx* ctx, void** image);
/* 1 when everything enqueued on the context's stream has completed, 0 when not yet,
 * < 0 on error (pairs with ythip_trace_samples_async / ythip_cancel). */
int ythip_poll(ythip_ctx* ctx);

/* ------------------------------------------------------------------------- */
/* The hot path                                                                */
/* ------------------------------------------------------------------------- */

/* trace_samples (yocto_trace.h:171-173, yocto_trace.cpp:1595-1619): renders
 * `params->batch` more samples for every pixel of the resident state slice and
 * bumps state.samples.  Returns immediately (no-op) when
 * state.samples >= params->samples.  `stop` (may be NULL) cancels the batch the way
 * trace_start's worker does (yocto_trace.cpp:1636-1637 checks context.stop before
 * every sample): the host watches `*stop` while the batch runs and relays it to a
 * device flag the kernel polls at every sample boundary, so the call returns within
 * about one sample's time with YTHIP_ERR_CANCELLED; as in the reference the pixels
 * have then taken different numbers of the batch's samples and state.samples is not
 * advanced.  Synchronous. */
int ythip_trace_samples(ythip_ctx* ctx, const ythip_params* params,
    const volatile int32_t* stop);
/* Same, but only enqueues the work on the stream (pair with ythip_sync). */
int ythip_trace_samples_async(ythip_ctx* ctx, const ythip_params* params);
/* trace_cancel (yocto_trace.h:219) for an enqueued batch: raises the device flag from
 * another host thread / after ythip_trace_samples_async; the batch winds down at its
 * pixels' sample boundaries.  The next batch lowers the flag again. */
int ythip_cancel(ythip_ctx* ctx);
/* trace_sample (yocto_trace.h:174-176, yocto_trace.cpp:1461-1492): ONE sample of the
 * pixel (i, j) of the frame — the pixel's next four rng draws, one path, the
 * running-mean update with weight 1 / (sample + 1), hits += 1.  Leaves
 * state.samples alone, as the reference does.  YTHIP_ERR_INVALID when the pixel is
 * not part of the resident slice.  Synchronous; a one-workgroup launch of the same
 * kernel trace_samples uses. */
int ythip_trace_sample(ythip_ctx* ctx, const ythip_params* params, int i, int j,
    int sample);

/* Scheduling inside the persistent kernel (no reference equivalent; results do
 * not depend on it).  adaptive_wait = 1 (default): a workgroup that has measured
 * its bounce rays to be much cheaper than its camera rays lets regenerated camera
 * rays wait until the pending bounce rays are done and traces them together
 * (DESIGN.md §4); 0: every queued ray runs in every iteration.  The environment
 * variable YTHIP_HOLD=0/1 sets the default of new contexts (A/B measurements). */
int ythip_set_scheduling(ythip_ctx* ctx, int adaptive_wait);
/* Early miss (no reference equivalent; results do not depend on it).  1 (default):
 * trace_path / trace_pathtest / trace_naive / trace_eyelight decide at the end of a bounce whether the next ray can
 * enter the scene's root box at all (the test intersect_scene_bvh opens with,
 * yocto_bvh.cpp:554-590) and, if not, take the next iteration's miss branch
 * (yocto_trace.cpp:473-477) in place; 0: every continuing ray goes through the queue.
 * YTHIP_PEEK=0/1 sets the default of new contexts (A/B measurements). */
int ythip_set_early_miss(ythip_ctx* ctx, int enable);

/* Kernel specialisation by scene content (results do not depend on it).  1 (default):
 * when every material of the resident scene is matte and untextured and every shape
 * a triangle mesh, the `path` sampler runs a k_trace variant compiled without the
 * other seven lobes, the volume and texture code and the quad / line / point paths
 * (10 % faster on BASELINE configs[1]); 0: always the general kernel. */
int ythip_set_specialization(ythip_ctx* ctx, int enable);

/* ---- the file formats either side of the path (host code, no device needed) ----
 * trace_params <-> JSON: the reference's parameter files (libs/yocto/yocto_sceneio.cpp:5815-5852,
 * load / save / update_trace_params :5933-5945): same keys, enums as their labels.
 * ythip_params_default = trace_params{} (yocto_trace.h:95-113).  ythip_params_from_json has the
 * reference's "update" semantics: a key that is absent keeps the value already in *params; `length`
 * < 0 = strlen(text).  ythip_params_to_json returns the length of the text (writes at most
 * `capacity` bytes incl. the terminator; call with NULL / 0 for the size), -1 on a bad enum value. */
void    ythip_params_default(ythip_params* params);
int     ythip_params_from_json(const char* text, int64_t length, ythip_params* params);
int64_t ythip_params_to_json(const ythip_params* params, char* buffer, int64_t capacity);
const char* ythip_io_last_error(void);
/* PLY -> flat pools, without the reference's ply_model -> shape_data -> flatten() generations of
 * copies (load_ply, libs/yocto/yocto_modelio.cpp:487-740; load_shape's getters,
 * libs/yocto/yocto_sceneio.cpp:1017-1033 with yocto_modelio.h:548-817).  ythip_ply_open maps the file
 * and reports in the num_* fields of `counts` what load_shape would produce from it (faces: quads if any
 * face has four corners, else triangles, polygons fanned; "empty shape" is an error, as there); the
 * caller sizes ythip_scene_staging from the counts of all its shapes and has ythip_ply_read convert the
 * properties straight into the pools at the shape's offsets (NULL = skip that array; colors get alpha
 * 1 when the file has none; flip_texcoord: v -> 1 - v, what load_shape is called with for scenes). */
typedef struct ythip_ply ythip_ply;
int  ythip_ply_open(const char* path, ythip_ply** ply, ythip_shape* counts);
int  ythip_ply_read(ythip_ply* ply, int flip_texcoord, float* positions, float* normals, float* texcoords,
    float* colors, float* radius, int32_t* points, int32_t* lines, int32_t* triangles, int32_t* quads);
void ythip_ply_close(ythip_ply* ply);

/* A scene FILE into the flat pools — the reference's load_scene for its builtin JSON format
 * (load_json_scene, libs/yocto/yocto_sceneio.cpp:3618-3857: scene.json, its PLY shapes, its HDR / PNG textures)
 * without scene_data's vector-of-vectors generation.  ythip_scene_open parses scene.json (versions 4.2 /
 * 5.0, and 4.0 = files without asset.version, load_json_scene_version40 :3025-3373), maps every PLY and reads every texture header, and reports all the num_* of ythip_scene in
 * `counts`; ythip_scene_read fills caller pools of those sizes (every pointer of `pools`, writable — the
 * pools of ythip_scene_staging, or plain memory): records with the reference's defaults and fix-ups
 * (lookat, add_missing_camera :2119-2139, add_missing_radius :2142-2148, load_texture's `linear`),
 * shapes converted by ythip_ply_read, textures (Radiance HDR, OpenEXR, PNG, JPEG, BMP, TGA — by content, as
 * stb_image sniffs it) decoded to what stbi_loadf / LoadEXR / stbi_load(…, 4) return,
 * shapes and textures on `threads` threads (<= 0: one per hardware thread).  The pools equal the
 * reference loader's scene_data flattened, byte for byte.  Not read here, refused by name: subdivs,
 * format 4.1, PLY instance files, non-PLY shapes, GIF / PSD / PIC / PNM content and .ypreset textures.
 * ythip_load_scene = open + ythip_scene_staging + read + ythip_upload_scene_staged (`staged`, optional,
 * receives the pools: pass it to ythip_build_bvh / ythip_build_lights).  ythip_scene_find_camera mirrors
 * find_camera (yocto_scene.cpp:656-675); ythip_scene_name: `what` 0 camera, 1 instance, 2 environment,
 * 3 shape, 4 texture, 5 material. */
typedef struct ythip_scene_file ythip_scene_file;
int         ythip_scene_open(const char* path, ythip_scene_file** file, ythip_scene* counts);
int         ythip_scene_read(ythip_scene_file* file, const ythip_scene* pools, int threads);
int32_t     ythip_scene_find_camera(const ythip_scene_file* file, const char* name);
const char* ythip_scene_name(const ythip_scene_file* file, int what, int32_t index);
void        ythip_scene_close(ythip_scene_file* file);
int         ythip_load_scene(ythip_ctx* ctx, const char* path, int threads, ythip_scene* staged);

/* The pixel pool: a batch launched as fewer workgroups than tiles (16 per CU) whose lanes, when their pixel has
 * taken its samples, take the next pixel of a queue (tiles in launch order) instead of idling until the tile is
 * done.  It fills the wavefronts of frames whose pixels cost very differently (BASELINE configs[4], hair: +15 %)
 * and costs a few per cent on even frames, so by default (mode 1) the library measures per trace state: once the
 * tile costs are known, one batch of >= 8 samples is timed plain, the next as a pool launch, and the faster per
 * sample is kept.  Mode 0 never, 2 always (env YTHIP_PIXEL_POOL).  The reference's results either way, bit for
 * bit: which lane traces a pixel's next sample is not observable (yocto_trace.cpp:1595-1619 is a parallel_for
 * over pixels). */
typedef struct ythip_pool_info {
  int32_t mode, workgroups, decided, on;
  float   plain_ms_per_sample, pool_ms_per_sample; /* the two timed batches, 0 until decided */
} ythip_pool_info;
int ythip_set_pixel_pool(ythip_ctx* ctx, int mode, int workgroups); /* workgroups <= 0: keep (default 16 per CU) */
int ythip_get_pixel_pool(ythip_ctx* ctx, ythip_pool_info* info);

/* The scheduler of trace_samples (round 6).  0: the fused persistent kernel — one wavefront owns a 16 x 4 pixel
 * tile for the whole batch (k_trace).  1: the STREAMING scheduler north_star describes — every pixel in flight, SoA ray /
 * hit / path state in HBM, and per bounce ("generation") a counting sort of the live paths' next rays by direction octant
 * and origin cell, a traversal-only extend kernel over the sorted queue, and a shade kernel in pixel order that accumulates,
 * regenerates and emits the next keys (csrc/yt_stream.h).  It schedules the SAME per-pixel operations — the parallel_for
 * over pixels of yocto_trace.cpp:1595-1619 says nothing about which worker runs a pixel — so the whole trace_state is the
 * megakernel's, bit for bit (tests/test_gpu_stream.py).  It serves the samplers `path`, `pathdirect` (the NEE ray of a bounce is
 * walked inside the shade stage, as k_trace's deferred stage walks it), `naive` and `pathtest` on scenes the wide walk serves, batches of
 * >= 4 samples, in every ythip_params::fastmath mode (0: the reference's bytes; 1 / 2: the bytes of that mode's fused kernel —
 * the tolerance and own-tree units carry their own build of the scheduler's kernels); anything else runs on the fused kernel
 * (ythip_get_stream_info says which ran, ythip_last_launch_fastmath which mode).
 * env YTHIP_SCHEDULER=0 / 1 / 2 sets the mode of new contexts (default 2: ythip_set_scheduler). */
typedef struct ythip_stream_info {
  int32_t ran;          /* the last batch ran on the streaming scheduler */
  int32_t generations;  /* generations that had rays queued */
  int32_t launched;     /* generations enqueued (the surplus returned at once) */
  int32_t bins;         /* bins of the counting sort (bounce-ray + camera-ray) */
  int32_t groups;       /* chains of generations that ran side by side */
  int32_t path_slots;   /* paths in flight: one per slot of the slice's tile grid */
  int64_t rays;         /* profiling (ythip_set_profiling bit 0) only: rays walked by ks_extend ... */
  int64_t lane_steps;   /* ... the traversal steps they took ... */
  int64_t wave_steps;   /* ... and 64 x the longest lane of every wavefront: lane_steps / wave_steps = how even the walks are */
  int64_t finish_rays;  /* queue entries handed to ks_finish (ythip_set_stream_finish): the paths that left the generations for the tail kernel */
  /* ythip_set_scheduler(ctx, 2), the measured choice: 0 a fused batch is timed next (once the fused path has settled: tile costs
   * known, pixel pool decided), 1 a streamed batch runs next untimed (the scheduler's buffers and first launches), 2 a streamed
   * batch is timed next, 3 waiting for the two times, 4 decided */
  int32_t choice_state;
  int32_t choice_streamed;      /* the decision (state 4): 1 = this state / sampler / mode / batch size is served streamed */
  float   fused_ms_per_sample;  /* the two timed batches, per sample per pixel (state 4) */
  float   stream_ms_per_sample;
} ythip_stream_info;
/* mode 0: the fused persistent kernel.  1: the streaming scheduler wherever it serves the batch.  2 (default): a MEASURED CHOICE,
 * as for the pixel pool — on a batch the streaming scheduler serves (and of >= 8 samples), once the fused path has settled, one
 * batch is timed fused, the next two run streamed (the second timed), and whichever took less time per sample (the streamed one by 3 % at least) serves
 * the state from then on; measured again for a new trace_state, sampler, mode, bounce limit or batch size.  The two schedulers
 * produce the same bytes, so the choice is invisible in the results.  A streamed batch is enqueued by a host loop that returns when
 * the batch is done.  Under mode 2 only ythip_trace_samples — whose caller waits anyway — is ever streamed, and
 * ythip_trace_samples_async keeps returning at once (its batches run fused and take no part in the choice); under mode 1 the async
 * call blocks for a batch the scheduler serves.  ythip_multi launches ranks that may stream from a thread each.  env YTHIP_SCHEDULER. */
int ythip_set_scheduler(ythip_ctx* ctx, int mode);
int ythip_get_scheduler(ythip_ctx* ctx); /* the mode set (0 for a null context) */
/* 1 when the next batch with these parameters may run streamed (its enqueue call then returns only when the batch is done), 0 when it
 * will run on the fused kernel; what ythip_multi asks before it launches its ranks from a thread each. */
int ythip_may_stream(ythip_ctx* ctx, const ythip_params* params);
/* Tuning of the streaming scheduler's sort (a negative argument keeps the current value; results never depend on it):
 * order 0 = direction octant major, origin cell minor, 1 = cell major (default), 2 = no sort (the queue in pixel order: the
 * baseline the sort is measured against); cell_bits 1..5 = the scene's root box cut into 2^bits cells per axis (default 3);
 * phased 0 / 1 = ks_extend's majority-phase scene walk off / on (default off: sorted wavefronts mostly want the same
 * step kind; the own-tree mode keeps the fused kernel's default, on for matte scenes with area lights).  env YTHIP_STREAM_ORDER / _CELLS / _PHASED. */
int ythip_set_stream_options(ythip_ctx* ctx, int order, int cell_bits, int phased);
/* groups 1..8 (default 2): the pixels of the slice as that many runs, each a chain of generations of its own on its own stream —
 * one run's shade / sort launches fill the machine while another's extend launch drains (frames too small for it run as
 * fewer).  env YTHIP_STREAM_GROUPS.  Results never depend on it. */
int ythip_set_stream_groups(ythip_ctx* ctx, int groups);
/* The tail of a streamed batch: once a group's queue has shrunk to `permille` thousandths of its path slots (default 250), the
 * group stops running generations and ONE launch (ks_finish: a lane per queue entry, extend and shade in turn on the same HBM
 * state) carries the paths still queued to the end of their pixels' batch — the last third of a batch's generations holds 2-4 %
 * of its rays.  0 = never (generations until the queue is empty), 1000 = from the first ray on (tests).  env YTHIP_STREAM_FINISH.
 * Results never depend on it. */
int ythip_set_stream_finish(ythip_ctx* ctx, int permille);
int ythip_get_stream_info(ythip_ctx* ctx, ythip_stream_info* info);
/* Profiling (ythip_set_profiling bit 0 during the batch): the queue length of every generation of the last streamed batch,
 * up to `capacity` (and 8192) entries; *written = how many. */
int ythip_get_stream_generations(ythip_ctx* ctx, int32_t* rays, int32_t capacity, int32_t* written);
/* Profiling: the traversal steps (+ 1) of every ray of ONE generation of a streamed batch, in queue order per group (a group's
 * entries start at its first path slot; entries beyond the generation's queue length keep what an earlier batch left).  Call with
 * steps = NULL before the batch to choose the generation (-1: off), with a buffer of >= path_slots entries after it. */
int ythip_get_stream_walk_steps(ythip_ctx* ctx, int generation, int32_t* steps, int32_t capacity);

/* The mode the last trace_samples / trace_sample launch of this context ran: 0 the bit-exact kernels, 1 the
 * tolerance-mode kernels, 2 the own-tree kernels (ythip_params::fastmath asked for it AND such a kernel exists for the
 * sampler and the resident scene). */
int ythip_last_launch_fastmath(ythip_ctx* ctx);

/* Which BVH walk k_trace's extend stage and the test entries below use: 0 the
 * binary walk (one sibling pair per dependent fetch), 1 the wide walk (the four
 * grandchildren per fetch: half the fetch chain, yt_bvh.h), 2 (default) chosen by
 * the resident BVH (wide unless every tree is tiny, < 64 primitives).  All give
 * the reference's hit records bit for bit; find_any queries, irregular rays and
 * the work-counting launches always walk binary. */
int ythip_set_traversal(ythip_ctx* ctx, int mode);

/* intersect_scene_bvh for a batch of rays (yocto_bvh.h:105-106,
 * yocto_bvh.cpp:554-617); parity/test entry using the same device function as
 * the extend kernel.  Host pointers. */
int ythip_intersect_batch(ythip_ctx* ctx, const ythip_ray* rays, int64_t n,
    int find_any, ythip_hit* hits);
/* intersect_instance_bvh for a batch (yocto_bvh.cpp:619-628). */
/* the same batch through the OWN tree's walk (instances may be NULL = intersect_scene; no find_any): a measuring
 * entry — tests/test_gpu_own_tree.py counts how often its records differ from ythip_intersect_batch's */
int ythip_intersect_batch_own(ythip_ctx* ctx, const int32_t* instances, const ythip_ray* rays, int64_t n,
                              ythip_hit* hits);
int ythip_intersect_instance_batch(ythip_ctx* ctx, const int32_t* instances,
    const ythip_ray* rays, int64_t n, int find_any, ythip_hit* hits);
/* The device's restatement of the reference platform's libm (yt_libm.h: glibc 2.35's sinf,
 * cosf, expf, exp2f, logf, atanf, acosf, atan2f, powf — fn 0..8; 9 fmodf, 10 sqrtf, 11 x / y)
 * evaluated on the device for n arguments (y may be NULL for the one-argument functions).
 * Test entry: the results must equal the host's glibc bit for bit. */
int ythip_test_libm(ythip_ctx* ctx, int fn, const float* x, const float* y, int64_t n, float* out);
/* Parity is pinned to ONE libm: glibc 2.35's x86-64 `_fma` variants (yt_libm.h).  This asks whether
 * the host's own libm — what a reference built on this machine renders with — agrees with the device
 * on 2,304 probe arguments: 1 yes, 0 no (the first disagreement in ythip_last_error: bit parity with a
 * reference run on THIS host is then not to be expected), < 0 error.  The drop-in shim calls it once
 * per process and warns on stderr. */
int ythip_host_libm_matches(ythip_ctx* ctx);
/* sample_camera for the next sample of every resident pixel
 * (yocto_trace.cpp:338-358,1467-1468) WITHOUT advancing the resident rngs;
 * writes width*(rows) rays.  Test entry. */
int ythip_camera_rays(ythip_ctx* ctx, const ythip_params* params,
    ythip_ray* rays);

/* ------------------------------------------------------------------------- */
/* One process, several GPUs (SURVEY.md §8(b) `ythip_create(device_ids[], n)`, §8(e)) */
/* ------------------------------------------------------------------------- */
/* A ythip_multi owns one context per device id (ids may repeat: several ranks on one GPU,
 * a rehearsal of the sharding).  Scene, BVH and lights are full replicas: upload / build them
 * on every rank's context (ythip_multi_ctx).  trace_state is sharded by 16-pixel tile
 * columns dealt round-robin (rank r of n: ythip_state_create_striped(.., r, n)); a pixel's
 * samples stay on one device (its PCG stream and running means are order-dependent), pixels
 * are independent (yocto_trace.cpp:1600-1612), so there is NO data-path collective — only the
 * framebuffer gather of ythip_multi_get_image: RCCL send/recv to device 0 over xGMI between
 * distinct devices, device copies when ranks share a device. */
typedef struct ythip_multi ythip_multi;
int         ythip_create_multi(const int* device_ids, int n, ythip_multi** out);
void        ythip_destroy_multi(ythip_multi* multi);
int         ythip_multi_size(const ythip_multi* multi);
ythip_ctx*  ythip_multi_ctx(ythip_multi* multi, int rank);
const char* ythip_multi_last_error(const ythip_multi* multi);
/* make_trace_state's storage for a width x height frame, one column-striped slice per rank */
int ythip_multi_state_create(ythip_multi* multi, int width, int height);
/* full-frame host arrays (reference layout, any may be NULL) <-> the ranks' slices */
int ythip_multi_state_upload(ythip_multi* multi, const float* image, const float* albedo,
    const float* normal, const int32_t* hits, const uint64_t* rngs, int samples);
int ythip_multi_state_download(ythip_multi* multi, float* image, float* albedo, float* normal,
    int32_t* hits, uint64_t* rngs, int* samples);
/* trace_samples on every device concurrently (one host thread, asynchronous launches);
 * `stop` as in ythip_trace_samples, relayed to every rank. */
int ythip_multi_trace_samples(ythip_multi* multi, const ythip_params* params,
    const volatile int32_t* stop);
/* get_image (yocto_trace.cpp:1694-1709) of the whole frame: the framebuffer gather. */
int ythip_multi_get_image(ythip_multi* multi, float* image);
/* How the last gather moved the slices ("rccl send/recv", "device copies", ...) and the
 * rank count RCCL's communicator reports (0 when RCCL was not used). */
int ythip_multi_gather_info(const ythip_multi* multi, char* mode, int mode_len, int* rccl_ranks);

/* ------------------------------------------------------------------------- */
/* Measurement                                                                 */
/* ------------------------------------------------------------------------- */
/* mode bit 0: time k_trace launches with hipEvents on the launch stream;
 * mode bit 1: count traversal work (nodes/prims/instances) in-kernel. */
int ythip_set_profiling(ythip_ctx* ctx, int mode);
int ythip_reset_stats(ythip_ctx* ctx);
int ythip_get_stats(ythip_ctx* ctx, ythip_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* YTHIP_H */
