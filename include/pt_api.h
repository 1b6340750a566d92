/* pt_api.h — C ABI of the MI355X path-tracing hot path.
 *
 * This is the drop-in boundary for the reference's PT hot path
 * (/root/reference = gillett-hernandez/rust-pathtracer @ 2024_08_07):
 *
 *   pt_render            replaces  Renderer::render            src/renderer/mod.rs:107-112
 *                                  TiledRenderer::render_sampled src/renderer/tiled.rs:279-542
 *                                  PathTracingIntegrator::color src/integrator/pt.rs:397-615
 *                                  random_walk                  src/integrator/utils.rs:152-376
 *   pt_intersect         replaces  World::hit                   src/world/mod.rs:166-168
 *                                  Accelerator::hit             src/accelerator/mod.rs:86-178
 *   pt_bsdf_sample       replaces  Material::generate_and_evaluate src/materials/mod.rs:67-74
 *   pt_bsdf_eval         replaces  Material::bsdf               src/materials/mod.rs:59-66
 *   pt_emission          replaces  Material::emission           src/materials/mod.rs:115-117
 *   pt_scene_create      takes the data the reference keeps in `World`
 *                        (src/world/mod.rs:18-28) flattened to plain arrays:
 *                        Instance/Aggregate (src/geometry/instance.rs:9-15,
 *                        src/geometry/mod.rs:17-122), Mesh (src/geometry/mesh.rs:243-253),
 *                        MaterialEnum (src/materials/mod.rs:275-283), Curve /
 *                        CurveWithCDF (math crate; built in src/parsing/curves.rs:298-372),
 *                        TexStack (src/texture.rs:236-264), EnvironmentMap
 *                        (src/world/environment.rs:6-27), ProjectiveCamera
 *                        (src/camera/projective_camera.rs:8-25).
 *
 * A per-sample FFI (SamplerIntegrator::color, src/integrator/mod.rs:125-140) is
 * useless for a GPU, so the replacement plugs in at Renderer granularity: a
 * Rust `impl Renderer for HipRenderer` flattens its `World` into a
 * pt_scene_desc once and calls pt_render per RenderSettings (INTEGRATION.md
 * shows that binding).
 *
 * Conventions: plain pointers and sizes only; the library copies everything it
 * needs out of the descriptors during pt_scene_create (caller keeps ownership);
 * every call returns PT_OK (0) or a nonzero pt_status and never throws or
 * aborts across the boundary; pt_last_error() returns a thread-local message;
 * a pt_scene may be used from one thread at a time; calls block until the
 * result is complete.  All floats are IEEE f32, matrices are row-major 4x4.
 *
 * The CPU oracle (oracle/, test infrastructure only) exports the same
 * signatures with the prefix ptref_ instead of pt_.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t pt_status;
enum {
    PT_OK = 0,
    PT_ERR_INVALID_ARGUMENT = 1,
    PT_ERR_NO_DEVICE = 2,       /* no HIP device / HIP runtime failure: the product path never falls back to CPU */
    PT_ERR_OUT_OF_MEMORY = 3,
    PT_ERR_UNSUPPORTED = 4,
    PT_ERR_DEVICE = 5
};

/* MaterialId (src/materials/mod.rs:22-27) packed into 32 bits: tag << 16 | index. */
enum { PT_TAG_MATERIAL = 0, PT_TAG_LIGHT = 1, PT_TAG_CAMERA = 2 };
#define PT_MATERIAL_ID(tag, index) ((((uint32_t)(tag)) << 16) | ((uint32_t)(index) & 0xffffu))
#define PT_MATERIAL_TAG(id) (((id) >> 16) & 0x3u)
#define PT_MATERIAL_INDEX(id) ((id) & 0xffffu)
#define PT_MATERIAL_NONE 0xffffffffu /* Instance::material_id == None (src/geometry/instance.rs:13) */

/* ---- spectral curves: math::curves::Curve as built by src/parsing/curves.rs:298-372 */
enum {
    PT_CURVE_LINEAR = 0,          /* Curve::Linear{signal,bounds,mode}: p0 = lower, p1 = upper, data = signal[count] */
    PT_CURVE_TABULATED = 1,       /* Curve::Tabulated{signal:(x,y)[],mode}: data = x0,y0,x1,y1,... (count pairs) */
    PT_CURVE_CAUCHY = 2,          /* Curve::Cauchy{a,b}: p0 = a, p1 = b */
    PT_CURVE_EXPONENTIAL = 3,     /* Curve::Exponential{(offset,sigma_l,sigma_r,multiplier)[]}: data = count 4-tuples */
    PT_CURVE_INV_EXPONENTIAL = 4, /* Curve::InverseExponential, same layout */
    PT_CURVE_BLACKBODY = 5,       /* Curve::Blackbody{temperature,boost}: p0 = temperature, p1 = boost */
    PT_CURVE_CONST = 6            /* Curve::Const(v): p0 = v */
};
enum { PT_INTERP_LINEAR = 0, PT_INTERP_NEAREST = 1, PT_INTERP_CUBIC = 2 };

typedef struct pt_curve {
    int32_t kind;
    int32_t mode;          /* PT_INTERP_* for LINEAR / TABULATED */
    float p0, p1;
    uint32_t data_offset;  /* in floats, into pt_scene_desc::curve_data */
    uint32_t data_count;   /* samples / pairs / 4-tuples */
} pt_curve;

/* ---- textures: src/texture.rs */
enum { PT_TEXTURE1 = 1, PT_TEXTURE4 = 4 };
typedef struct pt_texture_layer {
    int32_t kind;          /* PT_TEXTURE1 (src/texture.rs:122-141) or PT_TEXTURE4 (src/texture.rs:24-119) */
    int32_t curves[4];     /* curve indices; Texture1 uses curves[0] */
    int32_t width, height;
    uint64_t data_offset;  /* in floats, into texture_data; Texture1: w*h floats, Texture4: w*h*4 floats, row-major */
} pt_texture_layer;
typedef struct pt_texstack {  /* TexStack (src/texture.rs:236-264) = sum of layers */
    int32_t first_layer, layer_count;
} pt_texstack;

/* ---- materials: src/materials */
enum {
    PT_MATERIAL_LAMBERTIAN = 0,    /* src/materials/lambertian.rs */
    PT_MATERIAL_GGX = 1,           /* src/materials/ggx.rs */
    PT_MATERIAL_DIFFUSE_LIGHT = 2, /* src/materials/diffuse_light.rs */
    PT_MATERIAL_SHARP_LIGHT = 3,   /* src/materials/sharp_light.rs */
    PT_MATERIAL_PASSTHROUGH = 4    /* src/materials/passthrough.rs: the boundary of a medium; colour = curve_bounce */
};
enum { PT_SIDED_FORWARD = 0, PT_SIDED_REVERSE = 1, PT_SIDED_DUAL = 2 }; /* math::Sidedness */

typedef struct pt_material {
    int32_t kind;
    int32_t texstack;      /* Lambertian::texture */
    float alpha;           /* GGX::alpha */
    int32_t curve_eta, curve_eta_o, curve_kappa; /* GGX spectral IOR curves */
    int32_t curve_emit, curve_bounce;            /* DiffuseLight / SharpLight */
    float sharpness;       /* SharpLight: the value from the scene file; the library applies 1 + |s| (sharp_light.rs:26) */
    int32_t sidedness;
    int32_t outer_medium, inner_medium; /* MediumId (GGX, PassthroughFilter; ggx.rs:188-189, passthrough.rs:6-7): 0 = vacuum, k = mediums[k - 1] */
} pt_material;

/* ---- participating media: src/mediums (read by the medium-aware walk only, pt_render_desc::medium_aware) */
enum { PT_MEDIUM_HG = 0, PT_MEDIUM_RAYLEIGH = 1 };
typedef struct pt_medium {
    int32_t kind;
    int32_t curve_g, curve_sigma_a, curve_sigma_s; /* HenyeyGreensteinHomogeneous (hg.rs:16-24): g is stored + 1 */
    int32_t curve_ior;                             /* Rayleigh (rayleigh.rs:6-9) */
    float corrective_factor;
} pt_medium;

/* ---- geometry: src/geometry */
enum { PT_SHAPE_RECT = 0, PT_SHAPE_SPHERE = 1, PT_SHAPE_DISK = 2, PT_SHAPE_MESH = 3 };
enum { PT_AXIS_X = 0, PT_AXIS_Y = 1, PT_AXIS_Z = 2 };

typedef struct pt_mesh {          /* Mesh (src/geometry/mesh.rs:243-253) */
    uint32_t vertex_offset;       /* in vertices (xyz triples) into pt_scene_desc::vertices */
    uint32_t vertex_count;
    uint32_t index_offset;        /* in uint32, into indices; 3 per face, relative to vertex_offset */
    uint32_t face_count;
    int32_t normal_offset;        /* in normals (xyz triples), indexed like vertices; -1 = no shading normals */
    int32_t face_material_offset; /* into face_materials (packed MaterialId per face); -1 = Material(0) */
} pt_mesh;

typedef struct pt_instance {      /* Instance (src/geometry/instance.rs:9-15) + its Aggregate */
    int32_t kind;                 /* PT_SHAPE_* */
    int32_t has_transform;        /* Instance::transform.is_some() */
    uint32_t material;            /* packed MaterialId override or PT_MATERIAL_NONE */
    int32_t mesh;                 /* PT_SHAPE_MESH: index into meshes */
    float origin[3];              /* rect / sphere / disk */
    float size[2];                /* rect (src/geometry/rect.rs:16) */
    float radius;                 /* sphere / disk */
    int32_t axis;                 /* rect normal axis */
    int32_t two_sided;            /* rect / disk */
    float forward[16];            /* Transform3::forward, row-major */
    float reverse[16];            /* Transform3::reverse (inverse) */
} pt_instance;

/* ---- environment: src/world/environment.rs:6-27 */
enum { PT_ENV_CONSTANT = 0, PT_ENV_SUN = 1, PT_ENV_HDR = 2 };
typedef struct pt_environment {
    int32_t kind;
    float strength;
    int32_t curve;                /* Constant / Sun colour */
    float angular_diameter;       /* Sun */
    float sun_direction[3];       /* Sun */
    int32_t texstack;             /* HDR texture stack */
    float rotation_forward[16];   /* HDR */
    float rotation_reverse[16];
    int32_t importance_width;     /* HDR: baked importance map resolution; 0 = unbaked (uniform sampling) */
    int32_t importance_height;
    int32_t importance_luminance_curve; /* curve used as the luminance weight when baking (y_bar when -1) */
} pt_environment;

/* ---- camera: ProjectiveCamera (src/camera/projective_camera.rs:27-95, parse defaults src/parsing/cameras.rs:132-148)
 * or PanoramaCamera (src/camera/panorama_camera.rs:18-95) */
enum { PT_CAMERA_PROJECTIVE = 0, PT_CAMERA_PANORAMA = 1 };
typedef struct pt_camera {
    float look_from[3], look_at[3], v_up[3];
    float vfov;                   /* projective: degrees */
    float focal_distance;         /* projective */
    float aperture_diameter;      /* projective */
    int32_t kind;                 /* PT_CAMERA_* */
    float fov[2];                 /* panorama: horizontal (0, 360] and vertical (0, 180] field of view in degrees */
} pt_camera;

typedef struct pt_scene_desc {
    uint32_t curve_count;        const pt_curve* curves;
    size_t curve_data_count;     const float* curve_data;
    uint32_t layer_count;        const pt_texture_layer* layers;
    uint32_t texstack_count;     const pt_texstack* texstacks;
    size_t texture_data_count;   const float* texture_data;
    uint32_t material_count;     const pt_material* materials; /* index 0 must be the error material (src/parsing/mod.rs:438-455) */
    uint32_t mesh_count;         const pt_mesh* meshes;
    size_t vertex_count;         const float* vertices;        /* xyz */
    size_t index_count;          const uint32_t* indices;
    size_t normal_count;         const float* normals;         /* xyz */
    size_t face_material_count;  const uint32_t* face_materials;
    uint32_t instance_count;     const pt_instance* instances; /* InstanceId = position in this array */
    uint32_t camera_count;       const pt_camera* cameras;
    pt_environment environment;
    float env_sampling_probability; /* World::env_sampling_probability (src/world/mod.rs:26,170-176) */
    uint32_t medium_count;       const pt_medium* mediums;     /* World::mediums (MediumTable); at most 255 */
} pt_scene_desc;

/* Which shard renders tile t (t = index in the reference's tile order, tiled.rs:190-277; tiles_per_row = full tiles per film row):
 * tiles are dealt along diagonals, (column + row) mod N for the full tiles, so that every shard samples every column and every row
 * of the film — dealing t mod N gives a shard whole columns whenever N divides the tiles per row (1024 / 32 = 32 with 8 GPUs), and the
 * columns of an image are not equally expensive (measured: +-9 % between the 8 column shards of the Cornell box). */
#define PT_TILE_SHARD(tile, tiles_per_row, shard_count) (((tile) + (tile) / ((tiles_per_row) ? (tiles_per_row) : 1u)) % (shard_count))

/* ---- render settings: RenderSettings (src/parsing/config.rs:45-62) + PathTracingIntegrator (src/integrator/pt.rs:16-26) */
typedef struct pt_render_desc {
    uint32_t width, height;
    uint32_t spp;                /* min_samples */
    uint32_t min_bounces;        /* russian_roulette_start_index (src/integrator/utils.rs:267) */
    uint32_t max_bounces;
    uint32_t light_samples;
    uint32_t only_direct;
    float wavelength_lo, wavelength_hi; /* default BOUNDED_VISIBLE_RANGE = [380,750] (src/integrator/mod.rs:65-68) */
    uint32_t camera_index;
    uint64_t seed;
    uint32_t tile_width, tile_height;   /* RendererType::Tiled (src/parsing/config.rs:112-121); 0 = 32 */
    uint32_t shard_index, shard_count;  /* this call renders the tiles t with PT_TILE_SHARD(t, ...) == shard_index; others stay zero. 0/0 = whole film */
    uint32_t hero_wavelengths;   /* 1, or 4 for the hero-wavelength variant */
    uint32_t first_sample;       /* render samples [first_sample, first_sample + sample_count) of the spp; */
    uint32_t sample_count;       /* 0 = all spp. With a partial range the film holds the un-normalised running sum. */
    uint32_t phase_samples;      /* samples summed before their sum is added to the pixel: 0 = 10 (TiledRenderer, src/renderer/tiled.rs:347-361);
                                    >= spp = all of them, then one division (NaiveRenderer, src/renderer/naive.rs:82-103) */
    uint32_t medium_aware;       /* IntegratorType::PT { medium_aware } (src/parsing/config.rs:60-75): random_walk_medium instead of
                                    random_walk (src/integrator/pt.rs:447, utils.rs:708-1103); single wavelength only */
} pt_render_desc;

/* ---- engine tuning: the run-time switches of the HIP engine as one explicit struct.  None of them changes a result (the GPU tests
 * compare every one against the default bit for bit); they choose kernel forms, staging and batch sizes.  A scene takes its tuning
 * when it is created and keeps it: pt_scene_create reads the PT_AMD_* environment ONCE through pt_tuning_default (the variable named
 * with each field), pt_scene_create_tuned takes the struct from the caller — a Rust host sets these per scene, and a variable that
 * changes after the scene exists changes nothing.  (The reference has no counterpart: its renderer has no forms to choose.) */
enum {
    PT_TUNE_NO_LDS = 1u << 0,          /* PT_AMD_NO_LDS: the blob is read from HBM/L2, nothing staged */
    PT_TUNE_NO_CORE_LDS = 1u << 1,     /* PT_AMD_NO_CORE_LDS: a blob too big to stage whole is not staged by its core either */
    PT_TUNE_NO_PARK = 1u << 2,         /* PT_AMD_NO_PARK: walked meshes of a hybrid scene are walked in line */
    PT_TUNE_NO_LIVE_LIST = 1u << 3,    /* PT_AMD_NO_LIVE_LIST: the light-sample kernel reads every item of a segment, not the list of those with a live ray */
    PT_TUNE_EXACT_SLAB = 1u << 4,      /* PT_AMD_EXACT_SLAB: the six-division box test everywhere */
    PT_TUNE_NO_CULL = 1u << 5,         /* PT_AMD_NO_CULL: no culling by the closest hit, no early stop */
    PT_TUNE_NO_SWEEP = 1u << 6,        /* PT_AMD_NO_SWEEP: the BVH walk even where a sweep table exists */
    PT_TUNE_NO_MESH_SWEEP = 1u << 7,   /* PT_AMD_NO_MESH_SWEEP: walked meshes never swept */
    PT_TUNE_NO_KNOWN_LIGHT = 1u << 8,  /* PT_AMD_NO_KNOWN_LIGHT: a light-sample ray tests its bounding light again in phase 3 */
    PT_TUNE_GENERAL_FORMS = 1u << 9,   /* PT_AMD_GENERAL_FORMS: no kernel forms specialised by what the scene lacks (transforms) */
    PT_TUNE_NO_FUSE = 1u << 10,        /* PT_AMD_NO_FUSE: k_extend + k_shade as two launches even where the fused form (k_shade tracing its own segments) exists */
    PT_TUNE_NO_STAGE_TIMING = 1u << 11,/* PT_AMD_STAGE_TIMING=0: no HIP events around the launches (pt_profile::kernel_seconds stay 0) */
    PT_TUNE_MULTI_RCCL = 1u << 12,     /* PT_AMD_MULTI_RCCL: pt_render_multi takes the RCCL reduce even for one device */
    PT_TUNE_NO_ONE_LIGHT = 1u << 14,   /* PT_AMD_NO_ONE_LIGHT: in a scene with ONE light the lean vertex kernel does not run that light's shape test on its light-sample rays
                                          (by default a ray that misses the only light is dead where it is made and its item, if no ray of it lives, never read) */
    PT_TUNE_NO_CONVEX = 1u << 15,      /* PT_AMD_NO_CONVEX: no use of the host's convex-body certificates (a light-sample ray that leaves a closed convex mesh instance inward is
                                          dead where it is made, one that leaves it outward does not park at that mesh again, a path segment refracted into it ends that mesh's search at its first
                                          interior acceptance: pt_blob.h PT_INST_CONVEX_*, PT_PATH_INSIDE_MARK, round 6) */
    PT_TUNE_NO_MESH_SHORTCUTS = 1u << 16, /* PT_AMD_NO_MESH_SHORTCUTS: no mesh is decided without its triangle tests — a closed mesh's inner balls (a bounded light ray through one is blocked)
                                          and its 18-DOP slabs (a ray outside one misses) are not used; the searches they cut short run in full.  A diagnostic: the films are the same bit for bit */
    PT_TUNE_NO_AXIS_SCAN = 1u << 13    /* PT_AMD_NO_AXIS_SCAN: the parked kernels walk a ray that is parallel to an axis of its mesh like any other (by default the
                                          whole wave scans the mesh's leaves for it: such a ray passes most boxes, AABB::hit ignoring the axes its direction is zero along) */
};
typedef struct pt_tuning {
    uint32_t flags;               /* PT_TUNE_* */
    uint32_t batch_slots;         /* PT_AMD_BATCH: path slots per pass; 0 = 2^27 (2^26 with hero wavelengths) */
    uint32_t blocks_per_cu;       /* PT_AMD_BLOCKS_PER_CU: queue segments = workgroups per CU; 0 = 64 */
    uint32_t park_blocks_per_cu;  /* PT_AMD_PARK_BLOCKS_PER_CU: persistent workgroups per CU of the parked kernels with dynamic units; 0 = 4 */
    int32_t park_dynamic;         /* PT_AMD_PARK_DYNAMIC: 0 static segments, 1 units from a counter, -1 = by staging mode */
    uint32_t shade_form;          /* PT_AMD_SHADE_FORM: 0 = the leanest form the scene allows, 1 = at least NO_ENV, 2 = FULL */
    uint32_t lds_all_limit;       /* PT_AMD_LDS_ALL_LIMIT: largest blob staged whole, bytes; 0 = 24 KB */
    uint32_t multi_virtual;       /* PT_AMD_MULTI_VIRTUAL: pt_render_multi treats every device of the mask as this many (test mode:
                                     k host threads, streams and replicas per device, films summed on the device); 0 / 1 = off */
    uint32_t walk_evict_below;    /* PT_AMD_WALK_EVICT_BELOW: the parked kernels leave a resumed wave's mesh walks once fewer lanes than this are still
                                     walking (the rays park again and go on in a later wave); 1 = never, at most 64; 0 = the default */
    uint32_t walk_search_below;   /* PT_AMD_WALK_SEARCH_BELOW: a mesh walk's inner loop (box to box until the lane holds a leaf) ends once fewer lanes than
                                     this are still searching while others hold a leaf; 1 = never, at most 64; 0 = the default */
    uint32_t park_block;          /* PT_AMD_PARK_BLOCK: the parked kernels (scenes with walked meshes, one wavelength per path) of a scene whose blob is staged by
                                     its core only but fits 72 KB (the gem scene, C3: 66 KB) can run workgroups of 512 or 1024 threads that stage the WHOLE blob
                                     in LDS while the other kernels keep their staging mode.  0 = the measured default (the light-sample kernel at 512 — at 256 in a
                                     scene with a certified convex body, whose light rays seldom reach the mesh —, the closest-hit kernel at 256), 512 / 1024 = both kernels
                                     at that size, 256 = off */
    uint32_t light_prepass_max;   /* PT_AMD_LIGHT_PREPASS_MAX: a light-sample ray aimed at a light is bounded by the nearest hit among ALL lights before it is traced (one box
                                     test per light and ray: it buys the early stop at the first occluder).  A scene with more lights than this traces such a ray as a
                                     plain closest-hit search instead (test_bokeh.toml: 82 lights).  0 = the default (16); 0xffffffff = always bound */
    uint32_t top_evict_below;     /* PT_AMD_TOP_EVICT_BELOW: scenes without a sweep table (more than 64 instances) walk the top-level tree lane by lane; once fewer lanes
                                     of a wave than this are still walking, those leave with their place, are parked like a ray at a mesh and go on in a later wave of 64
                                     such rays.  1 = never, at most 64; 0 = the default (48 for a scene of more than 64 instances without a mesh, else never) */
    uint32_t group_evict_below;   /* PT_AMD_GROUP_EVICT_BELOW: the grouped sweep of a walked mesh of up to 384 triangles (the closest-hit kernel of a scene with ONE such mesh): a
                                     ray enters 5.7 of the gem's 51 groups on average and a wave waits for its busiest lane; once fewer lanes than this still have a group to
                                     do they leave with their groups and go on in a later wave of 64 such rays.  1 = never, at most 64; 0 = the default */
    uint32_t reserved[2];         /* must be 0 */
} pt_tuning;
/* The defaults, overridden by whatever PT_AMD_* variables the environment holds at the time of the call. */
void pt_tuning_default(pt_tuning* tuning);

typedef struct pt_profile {      /* Profile (src/profile.rs:2-8) + timing */
    uint64_t bounce_rays, shadow_rays, light_rays, camera_rays, env_hits;
    double seconds;              /* render loop only: the window of src/renderer/tiled.rs:294 -> :536 */
    double kernel_seconds[8];    /* HIP-event time per stage: generate, extend, shade, shadow, accumulate (one event per launch boundary: a launch's time includes
                                    the few microseconds of gap in front of it; the stream's non-kernel work — memsets — is charged to no stage); pt_render_multi adds
                                    [5] = host seconds spent on set-up before the render window (replicas, streams, device films,
                                    communicator: paid by the first call with a device set and film size, cached on the scene after),
                                    [6] = host seconds of the film reduce; [7] spare */
    uint64_t kernel_launches[8];
    uint64_t stage_items[8];     /* work items per stage summed over launches: paths generated, segments extended,
                                    vertices shaded, light-sample items traced, pixels accumulated; [5] = tracked mediums a fifth nesting level
                                    dropped (medium-aware walk; 0 = the walk is the reference's); [6] = threads per workgroup of the parked
                                    kernels when they ran in their big-workgroup form (pt_tuning::park_block), else 0; [7] = 1 when the camera
                                    vertices went through the path queue as lean records (ray + wavelength: 28 of 64 bytes), else 0 */
} pt_profile;

typedef struct pt_hit {          /* HitRecord (src/hittable.rs:7-16) */
    float t;
    float point[3];
    float normal[3];
    float uv[2];
    uint32_t material;           /* packed MaterialId */
    uint32_t instance;
    int32_t valid;               /* 0 = miss */
} pt_hit;

typedef struct pt_scene pt_scene;

pt_status pt_scene_create(const pt_scene_desc* desc, pt_scene** out);   /* tuning = pt_tuning_default */
pt_status pt_scene_create_tuned(const pt_scene_desc* desc, const pt_tuning* tuning, pt_scene** out);
void pt_scene_destroy(pt_scene* scene);
const char* pt_last_error(void);

/* Film: caller-allocated width*height*4 f32 (X,Y,Z,0), row-major, y = 0 is the
 * top row (src/tonemap/mod.rs:237-239), already divided by spp (tiled.rs:396-398). */
pt_status pt_render(pt_scene* scene, const pt_render_desc* desc, float* film_xyzw, pt_profile* profile);
/* Same, but `film_device` is device memory (HBM) of the current HIP device, written on `hip_stream`
 * (a hipStream_t, may be NULL); the call returns after the stream work is complete. */
pt_status pt_render_device(pt_scene* scene, const pt_render_desc* desc, void* film_device,
                           void* hip_stream, pt_profile* profile);

/* The whole node from one blocking call — what `impl Renderer for HipRenderer` calls (Renderer::render, src/renderer/mod.rs:107-112,
 * is one blocking call; SURVEY 8(b): the library owns devices and streams).  `device_mask`: bit d = HIP device d; 0 = every visible
 * device.  The library keeps one scene replica, one stream and one host thread per device, deals the film's 32x32 tiles to the
 * devices (PT_TILE_SHARD, as shard_index / shard_count do across processes), and sums the device films into the first device of the
 * mask with one RCCL reduce (a single-process communicator over xGMI; the shards are disjoint, so the sum is a gather and the film
 * is bit-identical to the one-device film).  `desc->shard_count` must be 0.  Counters of `profile` are summed over the devices,
 * `seconds` is the wall time of the call's render window, the per-stage device seconds are summed over the devices. */
pt_status pt_render_multi(pt_scene* scene, const pt_render_desc* desc, uint64_t device_mask, float* film_xyzw, pt_profile* profile);
/* HIP devices visible to the library (0 without a device). */
uint32_t pt_device_count(void);

/* Probes of the trait surface, used for parity tests of single stages. Host arrays in, host arrays out. */
pt_status pt_intersect(pt_scene* scene, size_t n, const float* origins, const float* directions, pt_hit* hits);
pt_status pt_bsdf_sample(pt_scene* scene, uint32_t material, size_t n, const float* lambda, const float* wi,
                         const float* sample2d, float* f, float* wo, float* pdf);
pt_status pt_bsdf_eval(pt_scene* scene, uint32_t material, size_t n, const float* lambda, const float* wi,
                       const float* wo, float* f, float* pdf);
pt_status pt_emission(pt_scene* scene, uint32_t material, size_t n, const float* lambda, const float* wi,
                      float* emission);
pt_status pt_curve_eval(pt_scene* scene, uint32_t curve, size_t n, const float* lambda, float* value);
/* The first stage of one camera sample — the film jitter of render_sampled (src/renderer/tiled.rs:369-375), the wavelength and the clamp of
 * PathTracingIntegrator::color (src/integrator/pt.rs:406-414) and Camera::sample_we / get_ray (src/camera/projective_camera.rs:101-120,
 * panorama_camera.rs:71-95) — for n (pixel id = y * width + x, sample index) pairs of the render `desc` describes (width, height, seed, wavelength
 * bounds, camera_index): origins and directions as 3 floats each, one wavelength per sample. */
pt_status pt_camera_samples(pt_scene* scene, const pt_render_desc* desc, size_t n, const uint32_t* pixel, const uint32_t* sample,
                            float* origins, float* directions, float* lambda);

/* ---- film output stage: output_film (src/renderer/mod.rs:24-80) = tonemap + colour space + files ---- */
enum { PT_TONEMAP_CLAMP = 0, PT_TONEMAP_REINHARD0 = 1, PT_TONEMAP_REINHARD1 = 2 };      /* src/parsing/tonemap.rs:8-31 */
enum { PT_COLORSPACE_SRGB = 0, PT_COLORSPACE_REC709 = 1, PT_COLORSPACE_REC2020 = 2 };   /* ColorSpaceSettings, src/parsing/config.rs */
typedef struct pt_output_desc {
    uint32_t width, height;
    int32_t tonemap;          /* PT_TONEMAP_* */
    int32_t luminance_only;   /* Reinhard0/1: false selects the per-channel x3 variants (src/parsing/tonemap.rs:80-104) */
    float exposure;           /* Clamp: stops (2^exposure), default 0 */
    float key_value;          /* Reinhard */
    float white_point;        /* Reinhard1 max_white */
    int32_t colorspace;       /* PT_COLORSPACE_* */
    float factor;             /* output_film's factor * premultiply (src/renderer/mod.rs:24-26) */
} pt_output_desc;
/* film_xyzw: host, width*height*4 f32.  rgba8: width*height*4 bytes = what write_to_files puts in the PNG (tonemapped,
 * converted to the colour space's primaries, OETF-encoded, ceil(255 v) clamped; src/tonemap/mod.rs:316-333).
 * linear_rgb (may be NULL): width*height*3 f32 = the EXR payload (factor * film in linear RGB of those primaries, :232-251). */
pt_status pt_output_film(const pt_output_desc* desc, const float* film_xyzw, uint8_t* rgba8, float* linear_rgb);
/* PNG (8-bit RGBA with gAMA and cHRM chunks, mod.rs:291-314) and OpenEXR (float RGB scanlines + chromaticities, :253-279). */
pt_status pt_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba8, int32_t colorspace);
pt_status pt_write_exr(const char* path, uint32_t width, uint32_t height, const float* linear_rgb, int32_t colorspace);

/* ---- film comparison: the three modes of src/bin/compare_exr.rs:39-52,70-170 on raw float4 images (XYZ films
 * or linear RGB), plus the per-channel statistics the parity tests quote. ---- */
enum { PT_COMPARE_ABSOLUTE = 0, PT_COMPARE_RMSE = 1, PT_COMPARE_RELATIVE = 2 };
typedef struct pt_compare_stats {
    double linf[4];        /* max |image - truth| per channel */
    double mean_abs[4];    /* mean |image - truth| per channel */
    double rmse;           /* sqrt(mean over pixels and channels of (image - truth)^2) */
    float pixel_min, pixel_max; /* range of the per-pixel value: RMSE mode = the "minmax" the reference prints (compare_exr.rs:108-110),
                                   other modes = largest channel of the output pixel */
    uint64_t nonfinite;    /* pixels with a NaN/inf channel in either input */
} pt_compare_stats;
/* image, truth, out: host, width*height*4 f32.  out (may be NULL):
 *   ABSOLUTE  |image - truth| per channel (compare_exr.rs:74-82)
 *   RMSE      per pixel sqrt(sum of squared channel differences / 4), mapped through the viridis gradient over [min, max]
 *             (compare_exr.rs:93-127); alpha = 1
 *   RELATIVE  |image - truth| / truth per channel, non-finite -> 0 (compare_exr.rs:150-161) */
pt_status pt_compare_films(uint32_t width, uint32_t height, const float* image, const float* truth, int32_t mode, float* out, pt_compare_stats* stats);

/* Library / device identification, e.g. "gfx950 ... 256 CUs". */
const char* pt_device_info(void);

#ifdef __cplusplus
}
#endif
#endif /* PT_API_H */
