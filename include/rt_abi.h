/* C ABI of the MI355X render path (librt_amd.so).
 *
 * This is the boundary a host program binds to replace the reference's CPU render loop
 *   pub fn render(world: Arc<World>, camera: Arc<Camera>) -> Canvas      (src/lib.rs:77-133)
 * and the recursive shading entry point under it
 *   World::color_at(&self, ray: &Ray, remaining: usize) -> Color          (src/scene/world.rs:198-212).
 * The host flattens its scene graph into the plain tables below (one depth-first pre-order node
 * array plus side tables), hands them over once, and then renders any number of frames.
 *
 * Conventions: every function returns 0 on success and a negative rt_status on failure, never
 * throws or aborts; rt_last_error() gives a thread-local message.  All pointers in rt_scene_desc
 * are host pointers that the library copies during rt_scene_create; the caller keeps ownership.
 * A scene handle may be used from one thread at a time.  All arithmetic on the device is IEEE f64
 * in the reference's evaluation order with no fused multiply-add contraction.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1, /* malformed table, null pointer, index out of range */
    RT_ERR_HIP = -2,     /* a HIP runtime call failed */
    RT_ERR_LIMIT = -3,   /* scene exceeds a documented limit (lights, nesting, depth) */
    RT_ERR_NO_DEVICE = -4
} rt_status;

/* Limits checked by rt_scene_create / rt_render*. */
#define RT_MAX_LIGHTS 16       /* point lights per scene */
#define RT_MAX_DEPTH 5         /* `remaining` of the first ray; the reference uses 5 (src/lib.rs:21) */
#define RT_MAX_XFORM_NEST 8    /* nested groups or CSG shapes with a non-identity transform */
#define RT_MAX_PATTERN_NEST 8  /* nested gradient/blended patterns */
#define RT_TILE 8              /* a wavefront renders an 8x8 pixel tile */
#define RT_MAX_CSG_NEST 8      /* CSG shapes nested inside one another */
#define RT_STATIC_MAX_TILES (1 << 19) /* most 8x8 tiles of one rt_render_device_static call: a 7680x4320
                                         frame has 518,400 */
#define RT_MAX_CSG_HITS 128    /* most intersections one ray can have with an outermost CSG shape:
                                  sphere, cube 2 each; cylinder, cone 4; plane, triangles 1 */

/* Node kinds: the shapes of src/geometry/{group,sphere,plane,cube,cylinder,cone,triangle,
 * smooth_triangle,csg}.rs.  An RT_CSG node has exactly two children: its left operand is the next
 * node, its right operand starts where the left one's subtree ends (src/geometry/csg.rs:15-25). */
enum { RT_GROUP = 0, RT_SPHERE, RT_PLANE, RT_CUBE, RT_CYLINDER, RT_CONE, RT_TRIANGLE, RT_SMOOTH_TRIANGLE, RT_CSG };
enum { RT_NODE_CASTS_SHADOW = 1, RT_NODE_CLOSED = 2,
       /* RT_CSG: the Operation (src/geometry/csg.rs:27-31) in bits 4-5 */
       RT_NODE_CSG_DIFFERENCE = 0 << 4, RT_NODE_CSG_INTERSECTION = 1 << 4, RT_NODE_CSG_UNION = 2 << 4,
       RT_NODE_CSG_OP_MASK = 3 << 4 };

/* One scene-graph node.  Nodes are stored in depth-first pre-order, children in the order the
 * reference keeps them in Group::children (src/geometry/group.rs:19), top-level objects in
 * World::objects order (src/scene/world.rs:16).  That order is what breaks ties between equal
 * intersection times, because the reference's sorts are stable. */
typedef struct rt_node {
    int32_t type;     /* RT_GROUP ... RT_CSG */
    int32_t flags;    /* RT_NODE_* */
    int32_t parent;   /* enclosing group, -1 for a World object */
    int32_t skip;     /* first node after this node's subtree (== index + 1 for a leaf) */
    int32_t material; /* leaves: index into materials; groups and CSG shapes: -1 (a CSG shape's own
                         material is never used: intersections carry the operand that was hit) */
    int32_t xform;    /* index into xforms, or -1 when the node's transform is the identity */
    int32_t data;     /* group, CSG: aabbs index; cylinder/cone: ranges index; triangles: tris index; else -1 */
    int32_t ctx;      /* nearest enclosing group or CSG shape whose transform is not the identity, or -1 */
} rt_node;

/* Inverse of a node transform, computed on the host by the reference's cofactor algorithm
 * (src/matrices/matrix.rs:201-226).  inv holds rows 0..2 of the 4x4 inverse (row 3 is exactly
 * 0 0 0 1 for affine transforms); inv_t holds the upper-left 3x3 of its transpose, row-major. */
typedef struct rt_xform {
    double inv[12];
    double inv_t[9];
    double reserved[3];
} rt_xform;

typedef struct rt_aabb { double min[3]; double max[3]; } rt_aabb;   /* Group::bounds, group.rs:196-208 */
typedef struct rt_range { double minimum; double maximum; } rt_range; /* cylinder.rs:22-23, cone.rs:22-23 */
typedef struct rt_tri { double p1[3]; double e1[3]; double e2[3]; } rt_tri; /* triangle.rs:23-28 */
/* Flat triangle: n1 is the precomputed face normal (triangle.rs:48), n2/n3 unused.
 * Smooth triangle: the three vertex normals (smooth_triangle.rs:28-30). */
typedef struct rt_tri_normals { double n1[3]; double n2[3]; double n3[3]; } rt_tri_normals;

/* Phong material (src/materials/phong.rs:10-19). */
typedef struct rt_material {
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    int32_t pattern; /* root of the pattern tree, index into patterns */
    int32_t reserved;
} rt_material;

/* Pattern kinds: src/patterns/{solid,striped,gradient,ring,checker,blended,perturbed,test_pattern}.rs */
enum { RT_PAT_SOLID = 0, RT_PAT_STRIPED, RT_PAT_GRADIENT, RT_PAT_RING, RT_PAT_CHECKER, RT_PAT_BLENDED,
       RT_PAT_PERTURBED, RT_PAT_TEST };
typedef struct rt_pattern {
    int32_t type;
    int32_t a;     /* former / delegate, index into patterns, -1 if none */
    int32_t b;     /* latter, -1 if none */
    int32_t xform; /* index into pattern_xforms (12 doubles each, rows 0..2 of the inverse), -1 = identity */
    double color[3]; /* solid */
    double scale;    /* perturbed */
} rt_pattern;

typedef struct rt_light { double position[3]; double intensity[3]; } rt_light; /* point_light.rs:9-19 */

typedef struct rt_scene_desc {
    int32_t n_nodes;          const rt_node* nodes;
    int32_t n_xforms;         const rt_xform* xforms;
    int32_t n_aabbs;          const rt_aabb* aabbs;
    int32_t n_ranges;         const rt_range* ranges;
    int32_t n_tris;           const rt_tri* tris;  const rt_tri_normals* tri_normals;
    int32_t n_materials;      const rt_material* materials;
    int32_t n_patterns;       const rt_pattern* patterns;
    int32_t n_pattern_xforms; const double* pattern_xforms;
    int32_t n_lights;         const rt_light* lights;
} rt_scene_desc;

/* Camera block: what Camera::new derives (src/scene/camera.rs:17-43) plus rows 0..2 of the
 * inverse view transform.  width/height are the canvas size, i.e. Camera::width()/height()
 * (camera.rs:68-74). */
typedef struct rt_camera {
    int32_t width, height;
    double half_width, half_height, pixel_size;
    double inv[12];
} rt_camera;

/* The image is split into row tiles of RT_TILE rows; tile t belongs to rank t % world_size
 * (cyclic, like the reference's cyclic pixel partition, src/lib.rs:94-111).  A rank's rows are
 * stored compactly: its k-th tile occupies rows [k*RT_TILE, (k+1)*RT_TILE) of its buffer. */
typedef struct rt_params {
    int32_t max_depth;  /* `remaining` for primary rays, 0..RT_MAX_DEPTH */
    int32_t rank;       /* 0 <= rank < world_size */
    int32_t world_size; /* >= 1 */
    int32_t reserved;
} rt_params;

typedef struct rt_stats {
    uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract; /* intersect_world equivalents */
    float kernel_ms; /* device time of the render kernel (HIP events on its stream) */
    float reserved;
} rt_stats;

typedef struct rt_scene rt_scene; /* opaque; owns the device copies of the tables */

int rt_device_count(void);
/* Copies the tables to `device` (a HIP device ordinal) after validating them. */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* What rt_scene_create spent on a scene (host clock, ms) and what it built.  No counterpart in the
 * reference, whose scene graph is walked as it is (src/geometry/group.rs:133-154). */
typedef struct rt_build_info {
    float validate_ms; /* checking the tables */
    float records_ms;  /* laying the node table out as traversal and shading records (cull boxes, sorting) */
    float upload_ms;   /* device allocations and copies */
    int32_t n_nodes, n_records, n_cull_records;
    int32_t reserved;
    uint64_t walk_bytes;  /* device bytes of the traversal records (what a scene walk reads) */
    uint64_t table_bytes; /* device bytes of every table of the scene, walk_bytes included */
} rt_build_info;
int rt_scene_build_info(const rt_scene* scene, rt_build_info* out);

/* Rows of the frame that `params->rank` renders (the compact buffer height). */
int64_t rt_local_rows(int32_t height, const rt_params* params);
/* Frame row stored at compact row `local_row` of `params->rank`. */
int64_t rt_global_row(int64_t local_row, const rt_params* params);

/* Replaces render (src/lib.rs:77-133) for this rank's rows.  d_rgb is DEVICE memory holding
 * rt_local_rows * width * 3 doubles, pixel (x, local_row) at (local_row*width + x)*3.  The launch
 * is asynchronous on `hip_stream` (a hipStream_t, may be NULL for the default stream). */
int rt_render_device(rt_scene* scene, const rt_camera* camera, const rt_params* params,
                     double* d_rgb, void* hip_stream);
/* No counterpart in the reference.  rt_render_device for callers that cannot allocate device
 * memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the
 * scratch a launch needs beyond the frame -- one set of ray counters and the tile dispatch order,
 * 8.1 MB -- is a fixed array inside the library instead of buffers a scene allocates on demand, so
 * the call allocates and frees nothing.  It renders the same pixels as rt_render_device.  The part
 * of the frame `params->rank` renders may have at most RT_STATIC_MAX_TILES 8x8 tiles (RT_ERR_LIMIT
 * beyond that).  Every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two
 * streams, two scenes, two threads) race; calls on different GPUs do not.  rt_scene_ray_counts
 * after it reports this call's rays until the next launch on the GPU. */
int rt_render_device_static(rt_scene* scene, const rt_camera* camera, const rt_params* params,
                            double* d_rgb, void* hip_stream);
/* Same as rt_render_device, but synchronous into HOST memory; fills stats when not NULL.  The
 * frame crosses PCIe once: straight into rgb_out when that is page-locked memory (rt_host_alloc,
 * hipHostMalloc, hipHostRegister), else through a ring of page-locked slots that host threads empty
 * into rgb_out while the next slot fills. */
int rt_render(rt_scene* scene, const rt_camera* camera, const rt_params* params, double* rgb_out,
              rt_stats* stats);
/* render followed by Canvas::clamp on every channel (src/window/canvas.rs:50-60: 255*c saturated to
 * [0, 255] and rounded up, NaN -> 0): the values Canvas::to_ppm prints (src/window/canvas.rs:62-94),
 * 3 bytes per pixel in the same pixel order, without the f64 frame ever leaving the device.
 * d_rgb8 is DEVICE memory of rt_local_rows * width * 3 bytes. */
int rt_render_device_u8(rt_scene* scene, const rt_camera* camera, const rt_params* params,
                        uint8_t* d_rgb8, void* hip_stream);
/* Same, but synchronous into HOST memory; fills stats when not NULL. */
int rt_render_u8(rt_scene* scene, const rt_camera* camera, const rt_params* params, uint8_t* rgb8_out,
                 rt_stats* stats);
/* render followed by Canvas::to_ppm (src/window/canvas.rs:62-126), both on the device: the PPM text that
 * main.rs writes to the output file (header, one image row per group of lines, lines broken
 * before they reach 70 characters).  text_out is HOST memory of `capacity` bytes; *length gets the
 * size of the text.  A frame needs at most 32 + height * (width * 12 + 1) bytes; when `capacity`
 * is too small the call fails with RT_ERR_LIMIT and *length holds the size needed.  Whole frames
 * only (world_size 1). */
int rt_render_ppm(rt_scene* scene, const rt_camera* camera, const rt_params* params, char* text_out,
                  uint64_t capacity, uint64_t* length, rt_stats* stats);
/* Canvas::to_ppm alone, on `device`, for width * height * 3 clamped bytes in host memory. */
int rt_ppm_encode(int device, const uint8_t* rgb8, int32_t width, int32_t height, char* text_out,
                  uint64_t capacity, uint64_t* length);
/* Replaces World::color_at (src/scene/world.rs:198-212) for n rays given as packed xyz origins
 * and directions in host memory; writes n*3 doubles to host memory. */
int rt_color_at(rt_scene* scene, int64_t n, const double* origins, const double* directions,
                int32_t remaining, double* rgb_out, rt_stats* stats);

/* Page-locked host memory, which the GPU writes directly: a frame buffer taken from here makes
 * rt_render, rt_render_u8, rt_render_ppm and rt_render_multi* skip the staging ring.  No counterpart
 * in the reference (its Canvas is ordinary memory, src/window/canvas.rs:8-12).  NULL on failure. */
void* rt_host_alloc(uint64_t bytes);
void rt_host_free(void* p);

/* After the stream has been synchronised: device times (ms) of the most recent launches on this
 * scene, oldest first, at most `cap`; returns how many were written. */
int rt_scene_launch_times(rt_scene* scene, float* ms, int cap);
/* Ray counters of the most recent launch (synchronises the launch's stream). */
int rt_scene_ray_counts(rt_scene* scene, rt_stats* stats);

/* ---- one frame on several GPUs from one call ------------------------------------------------------
 * render (src/lib.rs:77-133) is ONE call that fans the pixels out to its workers and collects them
 * again (lib.rs:85-126).  rt_multi does the same over the GPUs of one process: it owns a copy of the
 * scene on every GPU it uses, renders tile t (RT_TILE rows) on rank t % n_ranks, and assembles the
 * frame.  One call at a time per handle. */
typedef struct rt_multi rt_multi;
typedef struct rt_multi_stats {
    uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract; /* summed over the ranks */
    float kernel_ms;      /* longest render launch over the ranks (device time, HIP events) */
    float render_wall_ms; /* host clock: from the call to the end of the longest render (host variants: estimated
                             as kernel_ms after the call began; the first downloads overlap the last renders) */
    float exchange_ms;    /* host variants: from there until the last byte is in host memory; device variant:
                             device time of the gather on rank 0's GPU */
    float total_ms;       /* host clock around the whole call */
} rt_multi_stats;
/* devices: n_ranks HIP ordinals, or NULL for 0 .. n_ranks-1.  An ordinal may repeat: those ranks then
 * share a GPU (each with its own scene copy, stream and buffer), which is how the partition and the
 * assembly are exercised on a machine with one GPU. */
int rt_multi_create(const rt_scene_desc* desc, int32_t n_ranks, const int32_t* devices, rt_multi** out);
void rt_multi_destroy(rt_multi* multi);
int32_t rt_multi_ranks(const rt_multi* multi);
/* The whole frame into HOST memory (width * height * 3 doubles / bytes, row-major): every GPU copies
 * its own tiles over its own PCIe link to their rows of the frame. */
int rt_render_multi(rt_multi* multi, const rt_camera* camera, int32_t max_depth, double* rgb_out,
                    rt_multi_stats* stats);
int rt_render_multi_u8(rt_multi* multi, const rt_camera* camera, int32_t max_depth, uint8_t* rgb8_out,
                       rt_multi_stats* stats);
/* The whole frame into DEVICE memory on rank 0's GPU (u8 != 0: clamped bytes, else doubles): rank 0's
 * GPU pulls every peer's tiles over xGMI (peer-mapped reads, all links at once) and writes the rows in
 * frame order.  Synchronous. */
int rt_render_multi_device(rt_multi* multi, const rt_camera* camera, int32_t max_depth, void* d_frame,
                           int32_t u8, rt_multi_stats* stats);

/* Device self-test with no counterpart in the reference: the kernel divides by a ray direction
 * through its precomputed reciprocal (one exact correction step) instead of a full division per
 * box plane.  This entry evaluates both forms on the device for n host-supplied pairs so a test
 * can assert they agree bit for bit: out_fast[i] = corrected a[i] * (1/d[i]), out_ieee[i] = a[i]/d[i]. */
int rt_selftest_division(int device, int64_t n, const double* a, const double* d, double* out_fast,
                         double* out_ieee);

const char* rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
