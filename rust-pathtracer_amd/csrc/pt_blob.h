// pt_blob.h — layout of the flattened scene ("blob") the HIP kernels read.
//
// The whole scene except bulk texture texels is one array of 32-bit words, 16-byte aligned sections,
// addressed by word offsets from a header.  It has two parts: the core (header, curves, textures stacks, materials, mesh
// records, instances, top-level BVH, lights, sweep table) and, behind it, the mesh data (per mesh: BVH nodes, triangles,
// normals, leaf list) whose offsets are relative to the start of that part.  One layout, two homes: HBM, and — when it fits — a copy
// staged into LDS by every workgroup (Cornell: ~7 KB, brilliant-cut gem scene: ~45 KB), so BVH nodes,
// primitives, material records and spectral curve tables are read with ds_read instead of going to L2.
//
// Geometry keeps the reference's two-level structure (top-level BVH over Instances, per-mesh BVH over
// triangles, rays transformed into instance space; src/accelerator/mod.rs:86-178, src/geometry/mesh.rs:314-360,
// src/geometry/instance.rs:75-133) because hit distances, points and normals computed in instance space
// differ in the last bits from a world-space bake, and path decisions must match the oracle bit for bit.
#ifndef PT_BLOB_H
#define PT_BLOB_H
#include <stdint.h>

#define PT_BLOB_MAGIC 0x50544232u /* "PTB2": mesh records of 16 words, curve records with cell tables, instance records with their sweep-table bits (round 6) */
#define PT_NODE_INNER 0xffffffffu
#define PT_NODE_FLAT 0x80000000u  /* in the exit word [3]: the box has zero thickness along an axis */
#define PT_NODE_NO_CULL 0x40000000u /* in the exit word: the subtree holds a sphere, whose computed hit distance can fall short of its box by more
                                      than any relative margin (cancellation in the quadratic): never culled by the closest hit or a bound */
#define PT_NODE_CODE(w) (((w) >> 27) & 7u) /* in the exit word: the form of the filtered box test (0 thick, 1..3 flat along x / y / z, 4 flat along several axes) */
#define PT_NODE_EXIT(w) ((w) & 0x07ffffffu)

// A BVH node: two float4.  a = (min.xyz, bits(exit)), b = (max.xyz, bits(shape or PT_NODE_INNER)).
// Pre-order array with skip links (src/accelerator/lbvh.rs:16-45): an inner node that is hit continues at
// index + 1, otherwise (and after a leaf) at `exit`.  The reference's flat array has a "navigator" node in
// front of every leaf carrying the same box the leaf test recomputes; both are merged into the leaf here.
#define PT_NODE_WORDS 8

// Triangle: three float4: (p0.xyz, bits(material)), (p1.xyz, 0), (p2.xyz, 0); optional normals: three float4.
#define PT_TRI_WORDS 12
#define PT_TRI_FLAGS 11       /* the third vertex' spare word: the face's part of its instance's convex-body certificate (PT_INST_CONVEX_*, below; 0 = none) */
#define PT_TRI_IN_SAFE 1u     /* bit 0: a point of this face moved 1e-3 inward along any of the face's hit normals lies inside its closed convex body */
#define PT_TRI_IN_SAFE_INNER 2u /* bit 1: ... certified for the points of the face whose barycentric coordinates are all >= PT_TRI_INNER_BARY (a face at a sharp edge: a strip along that edge is not safe) */
#define PT_TRI_INNER_BARY 0.015625f
#define PT_TRI_OUT_SHIFT 2    /* bits 2..16: the face's OUTWARD threshold in units of 2^-15, rounded up: 0.02 + the sine of the largest angle between the face's own normal and its hit
                                 normals (vertex normals: a smooth-shaded mesh).  A ray that leaves a point of this face with n . d above it — n the hit normal — moves away from
                                 the face's plane, behind which the whole body lies.  0 = no such claim for this face. */
/* hit_record hands both on in the hit's instance word (Hit::instance) — only in a scene with a certificate, whose instances are numbered below 65536: */
#define PT_HIT_IN_SAFE 0x80000000u   /* bit 31: this hit may take the inward claim (PT_TRI_IN_SAFE, or _INNER and its barycentrics allow it) */
#define PT_HIT_OUT_SHIFT 16          /* bits 16..30: the face's outward threshold */
#define PT_HIT_INDEX_MASK 0xffffu

// Mesh record (8 words): node_off (float4 units... all offsets are WORD offsets), node_count, tri_off, normal_off
// (0 = none), face_count, leaf list, group boxes
#define PT_MESH_WORDS 16
#define PT_MESH_NODE_OFF 0
#define PT_MESH_NODE_COUNT 1
#define PT_MESH_TRI_OFF 2
#define PT_MESH_NORMAL_OFF 3
#define PT_MESH_FACE_COUNT 4
#define PT_MESH_LEAF_OFF 5     /* leaf list for mesh_sweep (0 = none): per leaf in pre-order 8 words = min.xyz, triangle word offset, max.xyz, flat */
#define PT_MESH_LEAF_COUNT 6
#define PT_MESH_GROUP_OFF 7    /* meshes of more than PT_MESH_GROUP_MIN leaves: per PT_MESH_GROUP consecutive leaves 8 words = the box that holds theirs (min.xyz, 0, max.xyz, flat); 0 = none */
/* A CLOSED mesh's inner sphere (round 5; mesh_surely_blocks, pt_device.h): centre [8..10] and radius [11] of a ball that lies strictly inside the surface (every
   edge shared by exactly two triangles, the centre at odd crossing parity, the radius 0.98 x the distance to the nearest triangle), in the mesh's own space; [12] the
   length of the bounding box's diagonal.  A ray that passes through the ball and leaves the box before its bound MUST cross the surface in between — the watertight
   triangle test cannot miss a crossing of a closed surface — so a light-sample ray's search can end there: "something opaque in front of the light", without a walk.
   Radius 0: no such ball (an open mesh, a light among its faces, no inside point found). */
#define PT_MESH_INNER_C 8
#define PT_MESH_INNER_R 11
#define PT_MESH_REACH 12
#define PT_MESH_MORE_OFF 13      /* further inner balls of the mesh (core section; four words each: centre, radius), 0 = none */
#define PT_MESH_MORE_COUNT 14
/* The mesh's bounding slabs along ten more directions — the six face diagonals (1, +-1, 0), (1, 0, +-1), (0, 1, +-1) and the four body diagonals (1, +-1, +-1): with the
   box a 26-DOP — in the mesh's own space (round 5; mesh_surely_missed, pt_device.h): [15] = offset of 21 words in the core section: per direction lo, hi (already widened by
   a margin of 1e-3 of the slab's width, rounded outward), then `far`: beyond this |n . o| the test is not trusted.  A ray that enters the mesh's box but misses one of
   these slabs cannot hit a triangle: no park, no walk (44 % of the rays through the brilliant cut's box miss the gem; the slabs know it for 37 %).  0 = none. */
#define PT_MESH_DOP_OFF 15
#define PT_MESH_DOP_DIRS 10
#ifndef PT_MESH_MORE_BALLS
#define PT_MESH_MORE_BALLS 7u
#endif
#ifndef PT_MESH_MORE_MIN
#define PT_MESH_MORE_MIN 0.25   /* a further ball is at least this fraction of the first one's radius */
#endif
#ifndef PT_MESH_GROUP
#define PT_MESH_GROUP 6   /* (measured on C3's gem, 302 leaves: k_extend_parked 4400 / 4190 / 3875 / 3605 / 3520 us at 16 / 12 / 8 / 6 / 5 leaves per group) */
#endif
#define PT_MESH_GROUP_MIN 96
#ifndef PT_MESH_SWEEP_MAX
#define PT_MESH_SWEEP_MAX 384  /* meshes of at most this many triangles get a leaf list (64 groups, one bit each) */
#endif

// Instance record (40 words).
#define PT_INST_WORDS 40
#define PT_INST_KIND 0
#define PT_INST_FLAGS 1      /* bit0 has_transform, bit1 two_sided, bits 2-3 axis, bits 4-5 PT_INST_CONVEX_* */
/* A mesh instance the host has CERTIFIED (pt_scene_host.cpp convex_certificate, f64, in world space) as a closed convex body whose hit normals are its faces' own
   (round 6; stage_shade, pt_stages.h).  A light-sample ray is made at a surface point p, offset by 1e-3 along the hit normal to the side it leaves on (pt.rs:176, 256):
   CONVEX_OUT: a ray that leaves such a body OUTWARD (n . d above the face's threshold, PT_TRI_OUT_SHIFT) starts outside the supporting plane of the face it left and moves
     away from it — the whole body lies behind that plane (to within 2e-4, checked), so no triangle of this instance can be hit: the light-sample kernel drops the instance from
     the ray's leaf mask (PT_INST_SWEEP_MASK) — no park, no walk.  (Optional: a traversal form that ignores the mark walks the mesh and finds nothing.)
   CONVEX_IN: a ray that leaves it INWARD starts at least 1e-4 inside every face plane (checked for every face), and every light lies outside the body's box (checked): the
     ray must cross the closed surface before it can meet a light, the reference's closest hit is that crossing (or something else in front of it — in any case no light,
     pt.rs:177-189), the sample contributes 0: the ray is dead where it is made (counted, never traced). */
#define PT_INST_CONVEX_OUT 16u
#define PT_INST_CONVEX_IN 32u
#define PT_INST_MATERIAL 2   /* packed MaterialId or PT_MATERIAL_NONE */
#define PT_INST_MESH 3       /* word offset of the mesh record */
#define PT_INST_ORIGIN 4     /* 3 floats */
#define PT_INST_RADIUS 7
#define PT_INST_SIZE 8       /* 2 floats */
#define PT_INST_SWEEP_MASK 10 /* 2 words: the instance's bits in the leaf sweep table (its own and its triangle leaves'); 0 = no table */
#define PT_INST_FORWARD 16   /* 12 floats: rows 0..2 of the 4x4 */
#define PT_INST_REVERSE 28   /* 12 floats */

// Material record (12 words).
#define PT_MAT_WORDS 12
#define PT_MAT_KIND 0
#define PT_MAT_TEXSTACK 1    /* word offset of texstack record */
#define PT_MAT_ALPHA 2
#define PT_MAT_ETA 3         /* word offsets of curve records */
#define PT_MAT_ETA_O 4
#define PT_MAT_KAPPA 5
#define PT_MAT_EMIT 6
#define PT_MAT_BOUNCE 7
#define PT_MAT_SHARPNESS 8   /* already 1 + |s| */
#define PT_MAT_SIDEDNESS 9
#define PT_MAT_METALLIC 10
#define PT_MAT_MEDIUMS 11    /* outer medium id | inner medium id << 8 (MediumId: 0 = vacuum, k = medium record k - 1) */

// Medium record (8 words): kind, curve record offsets of g / sigma_a / sigma_s (HG) and of the ior (Rayleigh), corrective factor, pad, pad
#define PT_MEDIUM_WORDS 8
#define PT_MED_KIND 0
#define PT_MED_G 1
#define PT_MED_SIGMA_A 2
#define PT_MED_SIGMA_S 3
#define PT_MED_IOR 4
#define PT_MED_CORRECTIVE 5

// Curve record (8 words): kind, mode, p0, p1, data_off (word offset), data_count, cell table, 1 / cell width
#define PT_CURVE_WORDS 8
// (round 5) a tabulated curve of 8..255 sorted knots: word 6 = (cells - 1) << 24 | word offset of its cell table (a byte per cell, four to a word: the number of
// knots in lower cells), word 7 = cells / (last knot - first knot) as f32; word 6 = 0: no table, the binary search
#define PT_CURVE_GRID 6
#define PT_CURVE_GRID_INV 7
#define PT_CURVE_GRID_MIN_KNOTS 8u
#define PT_BLOB_LDS_ALL_BYTES 24576u   /* the largest blob the kernels stage whole in LDS (pt_launch.h kLdsAllLimitBytes) */
// Texstack record: layer_count, then per layer 8 words: kind, curve0..3 (word offsets), width, height, texel offset (floats, into texture memory)
#define PT_LAYER_WORDS 8

// Header (72 words)
#define PT_HDR_WORDS 72
#define PT_HDR_MEDIUM_OFF 64       /* medium records (0 = none) */
#define PT_HDR_MEDIUM_COUNT 65
#define PT_HDR_MAGIC 0
#define PT_HDR_TOTAL_WORDS 1
#define PT_HDR_TOP_NODE_OFF 2
#define PT_HDR_TOP_NODE_COUNT 3
#define PT_HDR_INSTANCE_OFF 4
#define PT_HDR_INSTANCE_COUNT 5
#define PT_HDR_MATERIAL_OFF 6
#define PT_HDR_MATERIAL_COUNT 7
#define PT_HDR_LIGHT_OFF 8        /* u32 instance ids */
#define PT_HDR_LIGHT_COUNT 9
#define PT_HDR_ENV_KIND 10
#define PT_HDR_ENV_STRENGTH 11
#define PT_HDR_ENV_CURVE 12       /* word offset */
#define PT_HDR_ENV_ANGULAR 13
#define PT_HDR_ENV_SUN_DIR 14     /* 3 floats */
#define PT_HDR_ENV_PROB 17        /* get_env_sampling_probability() */
#define PT_HDR_WORLD_RADIUS 18
#define PT_HDR_CURVE_OFF 19
#define PT_HDR_CURVE_COUNT 20
#define PT_HDR_FLAGS 21
#define PT_HDR_LIGHT_NODE_OFF 22
#define PT_HDR_ENV_TEXSTACK 23     /* HDR: word offset of the texstack record */
#define PT_HDR_IMAP_ROWS 24        /* HDR importance map (0 = unbaked): rows, columns, float offsets into texture memory */
#define PT_HDR_IMAP_COLS 25
#define PT_HDR_IMAP_ROW_PDF 26
#define PT_HDR_IMAP_ROW_CMF 27
#define PT_HDR_IMAP_MARG_PDF 28
#define PT_HDR_IMAP_MARG_CMF 29
#define PT_HDR_SWEEP_OFF 30        /* leaf sweep table (0 = none): scenes with <= 64 leaves, see world_hit_sweep */
#define PT_HDR_SWEEP_COUNT 31      /* instances in top-level pre-order */
#define PT_HDR_ENV_FORWARD 32      /* 12 floats: rows 0..2 of the rotation */
#define PT_HDR_ENV_REVERSE 44      /* 12 floats */  /* per light: word offset of its top-level leaf node (its box gates the instance test) */
#define PT_FLAG_EXACT_SLAB 2u    /* diagnostics (PT_AMD_EXACT_SLAB=1): always take the six-division slab test */
#define PT_FLAG_NO_CULL 4u       /* diagnostics (PT_AMD_NO_CULL=1): never cull by the closest hit */
#define PT_FLAG_NO_SHADOW_BOUND 8u /* a mesh instance can produce a Light-tagged hit: the light pre-pass of shadow rays is off */
#define PT_FLAG_SWEEP_WALKS 32u   /* the sweep table holds mesh instances whose BVH is walked (hybrid form) */
#define PT_FLAG_NO_MESH_SWEEP 64u  /* diagnostics (PT_AMD_NO_MESH_SWEEP=1): walk every mesh BVH */
#define PT_FLAG_NO_SWEEP 16u      /* diagnostics (PT_AMD_NO_SWEEP=1): always walk the BVHs */
#define PT_FLAG_NO_KNOWN_LIGHT 256u /* diagnostics (PT_AMD_NO_KNOWN_LIGHT=1): phase 3 tests the nearest light again instead of taking the light pre-pass' distance */
#define PT_FLAG_NO_ONE_LIGHT 512u /* diagnostics (PT_AMD_NO_ONE_LIGHT=1): the lean vertex kernel does not test a light-sample ray against the scene's only light (stage_shade) */
#define PT_FLAG_NO_LIGHT_PREPASS 1024u /* the light list is long (pt_tuning::light_prepass_max): a light-sample ray is traced as a plain closest-hit search, without the
                                         pre-pass over every light's box that bounds it (nearest_light_hit is linear in the lights: 82 box tests per ray in test_bokeh.toml) */
#define PT_FLAG_CONVEX 2048u      /* some instance carries a PT_INST_CONVEX_* certificate (the vertex code looks at instance flags only then) */
#define PT_FLAG_REPLAY 128u       /* diagnostics (host emulation): phase 3 of the sweep as unbounded tests + ordered replay (the pooled form's logic) */
#define PT_FLAG_NO_TOP_CULL 1u   /* a Disk instance exists: its reference box (radius/2, disk.rs:24-28) does not contain it */

// Leaf sweep table (world_hit_sweep).  One mask bit per top-level leaf (instance) and per triangle leaf, numbered in
// traversal pre-order: instance j gets bit `first`, its triangle leaves first+1 .. first+count.
// Per instance (16 words; the mask bits are numbered in the pre-order of the top-level BVH's leaves, the records stand simple ones first, PT_HDR_SWEEP_SIMPLE):
//   [0] instance record offset, [1] kind | flat << 8 | has_transform << 9 | walked << 10 | form of the box test << 11 (0 thick, 1 / 2 / 3 flat
//   along x / y / z, 4 flat along several axes) | first mask bit << 16 | tested triangle leaves << 24,
//   [2..3] the mask its box test sets, [4..6] box min, [7] triangle-leaf list offset, [8..10] box max, [11] triangle-leaf count (all of them),
//   [12] the tested triangle leaves come grouped by the form of their box test: sizes of the groups 0..3, 8 bits each (the rest is group 4), [14] instance id
// Triangle leaf (8 words, grouped by test form and in the pre-order of the mesh BVH's leaves inside a group; only the leaves that keep a box test of their own): [0..2] box
//   min, [4..6] box max, [3] and [7] the low and high word of the mask the test sets = the leaf's own bit (the lowest) and the bits
//   of the later leaves of the instance whose box is bit-identical: they are tested against the same ray, so the decision is theirs
//   too and they have no record here.  Leaves whose box is the untransformed instance's own box ride in the instance's mask (the
//   two triangles of every planar quad share a box; the two of an axis-aligned wall share it with their instance).
// Bit table (PT_HDR_SWEEP_BITS_OFF): per mask bit 8 words: instance record offset, triangle word offset (0: the instance
// itself), box word offset (min at +0, max at +4), kind | flat << 8 | has_transform << 9 | instance id << 16,
// followers (2 words: the later bits that alias this one, directly or through a chain), then for a triangle the word
// distances from its record to its copies permuted for a dominant x axis (y, z, x) and y axis (z, x, y) — the mesh-data
// section keeps the triangles of every mesh in the table in all three vertex permutations (triangle_test_permuted)
#define PT_SWEEP_WALKED 0x400u     /* kind/flags word: a mesh instance whose triangles are not in the table; its BVH is walked */
#define PT_SWEEP_INST_WORDS 16
#define PT_SWEEP_TRI_WORDS 8
#define PT_SWEEP_BIT_WORDS 8
#define PT_SWEEP_MAX_BITS 64
#define PT_HDR_SWEEP_BITS_OFF 56
#define PT_HDR_IMAP_MARG_GUIDE 59   /* guide tables for the CDF searches (float offsets into texture memory, u32 bit patterns): */
#define PT_HDR_IMAP_ROW_GUIDE 60    /* n + 3 entries per table, entry j = first index whose cmf is >= j / n (pt_device.h sample_cmf) */
#define PT_HDR_IMAP_STRIDE 66        /* floats between consecutive entries of a pdf or cmf table: 2 = the tables are interleaved, (cmf[k], pdf[k]) pairs — the three or four
                                       values a sample ends on (cmf[k - 1], cmf[k], pdf[k], pdf[k + 1]) then lie on one 128-byte line instead of two */
#define PT_HDR_SWEEP_SIMPLE 67       /* 2 words: the sweep table begins with the instances whose box test is all there is to do, grouped by the form of the test:
                                       sizes of the groups 0..3 (8 bits each), then of group 4; the other instances follow (PT_HDR_SWEEP_COUNT counts all) */
/* ... and only while blob + tables leave the FULL vertex form its four workgroups per CU (four waves per SIMD, 160 KB of LDS: 40 KB each); a bigger staged
   blob keeps the tables in L2 rather than lose a workgroup per CU silently (round-4 advisor).  stage_marginal (device) and marginal_lds_bytes (host) share the rule. */
#define PT_SHADE_LDS_BUDGET 40960u
#define PT_MARG_LDS_BYTES(rows, has_guide) ((2u * (rows) + ((has_guide) ? (rows) + 3u : 0u)) * 4u + 16u)
#define PT_MARG_LDS_MAX_ROWS 2048u   /* importance maps of at most this many rows have their marginal tables staged in LDS by the FULL vertex form (24 KB + guide) */
#define PT_SPHERE_CULL_K 5.4e-3   /* 3.5 x 1.55e-3: see beyond_sphere, pt_device.h */
#define PT_HDR_TOP_MARGIN 70         /* word offset of one float per top-level BVH node (0 = none; round 6): 0 for a node that holds no sphere — it is culled by the closest hit
                                       with the plain margin —, else the constant part of the margin a node that holds spheres is culled with (PT_SPHERE_CULL_K, pt_device.h:
                                       3.5 r_max + K (diagonal + r_max), r_max the largest radius below the node); +inf = never (a transformed sphere below it) */
#define PT_HDR_CONVEX_INST 69        /* 1 + the id of the scene's ONLY instance with PT_INST_CONVEX_OUT (0: none, or several): a path segment that leaves it outward carries a mark
                                       — the sign of its record's previous-pdf word, which every reader squares — and the parked closest-hit kernel drops the instance from that
                                       ray's leaf mask, as the light-sample kernel does for marked light rays */
#define PT_PATH_INSIDE_MARK 0x80000000u /* in a path record's slot word (slots are numbered below 2^30): the segment starts INSIDE the scene's one certified convex body (it left an
                                          inward-safe face inward) — the parked closest-hit kernel ends that body's sweep at the first triangle accepted well inside itself (mesh_walk) */
#define PT_HDR_CORE_WORDS 61        /* words of the core section; the mesh-data section follows it */
#define PT_HDR_SWEEP_MESH_MASK 57  /* 2 words: the bits that stand for mesh instances (no primitive of their own) */
#define PT_HDR_SWEEP_OWNER_MASK 62 /* 2 words: the bits whose primitive test needs the ray itself (analytic shapes, triangles of transformed
                                      instances); the other bits are triangles tested with the world ray's shear constants (pooled phase 3) */

#endif
