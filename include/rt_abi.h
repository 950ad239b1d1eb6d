/*
 * rt_abi.h — C ABI of libraytrace_hip.so, the MI355X-native replacement for the
 * host→kernel call surface of SebLague/Ray-Tracing's compute path tracer.
 *
 * Every entry point below replaces a group of Unity ComputeShader / ComputeBuffer
 * calls made by the reference dispatcher `Assets/Scripts/Tracer/RayComputeManager.cs`
 * (RCM) on the kernels of `Assets/Scripts/Tracer/RayCompute.compute` (RCC) +
 * `RayCommon.hlsl` (RC).  The cited file:line is the reference interface replaced.
 *
 * Conventions
 *  - plain C, little-endian, 4-byte scalars, no padding inside the PODs (checked
 *    by static asserts below and by the abi_size handshake in RtParams);
 *  - matrices are Unity `Matrix4x4` memory order = column-major: m[c*4 + r];
 *  - images are RGBA32F, W*H*16 bytes, row 0 = BOTTOM of the image (RC:555);
 *  - every function returns RT_OK (0) or a negative RtStatus; text via rt_last_error;
 *  - one context is externally synchronised (one caller thread at a time);
 *  - the caller owns all host arrays (copied during the call).
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

typedef enum RtStatus {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1,   /* null pointer, negative count, bad size        */
    RT_ERR_ABI_MISMATCH = -2,  /* RtParams.abi_version / struct_size handshake   */
    RT_ERR_NO_DEVICE = -3,     /* no HIP device / HIP runtime failure at create  */
    RT_ERR_HIP = -4,           /* a HIP call failed (message in rt_last_error)   */
    RT_ERR_STATE = -5,         /* call order wrong (render before resize/scene)  */
    RT_ERR_SCENE = -6,         /* scene buffers inconsistent (index out of range,
                                  BVH deeper than RT_MAX_BVH_DEPTH, ...)         */
    RT_ERR_OOM = -7
} RtStatus;

/* Material flags — RayTracingMaterial.cs:7-12, RC:29-30 */
enum { RT_MATERIAL_DEFAULT = 0, RT_MATERIAL_CHECKERED = 1, RT_MATERIAL_GLASS = 2 };

/* BVH.cs:11-16 */
enum { RT_BVH_QUALITY_LOW = 0, RT_BVH_QUALITY_HIGH = 1, RT_BVH_QUALITY_DISABLED = 2 };

/* Deepest leaf the reference builder can emit (BVH.cs:91,101). The reference's
 * traversal stack has 32 ints (RC:239) which a depth-32 leaf overflows; ours holds
 * RT_MAX_BVH_DEPTH+2 entries so the result is defined for every tree the builder
 * can produce.  rt_upload_scene rejects deeper trees with RT_ERR_SCENE. */
#define RT_MAX_BVH_DEPTH 32

/* RC:64-76 == RayTracingMaterial.cs:15-27 (88 bytes) */
typedef struct RtMaterial {
    float diffuseCol[4];
    float emissionCol[4];
    float specularCol[4];
    float absorption[4];
    float absorptionStrength; /* C# name: absorptionMultiplier */
    float emissionStrength;
    float smoothness;
    float specularProbability;
    float ior;
    int32_t flag;
} RtMaterial;

/* RC:78-85 == RCM:256-263 nested MeshInfo (224 bytes) */
typedef struct RtModel {
    int32_t nodeOffset;
    int32_t triOffset;
    float worldToLocal[16]; /* column-major */
    float localToWorld[16]; /* column-major */
    RtMaterial material;
} RtModel;

/* RC:49-53 == BVH.cs:579-598 (72 bytes), mesh-local space */
typedef struct RtTriangle {
    float posA[3], posB[3], posC[3];
    float normA[3], normB[3], normC[3];
} RtTriangle;

/* RC:87-95 == BVH.cs:432-457 (32 bytes). Leaf iff triangleCount > 0 (RC:246). */
typedef struct RtBVHNode {
    float boundsMin[3];
    float boundsMax[3];
    int32_t startIndex;
    int32_t triangleCount;
} RtBVHNode;

/* Analytic sphere (104 bytes). NOT a buffer of the reference snapshot: the
 * reference keeps the maths (RaySphere, RC:289-332) but its only call is
 * commented out (RC:341).  BASELINE.json's configs use spheres, so the buffer is
 * an additive extension hooked where the commented call sits: spheres are tested
 * before the model loop, closest hit carried in result.dst, material taken from
 * the sphere instead of the hard-coded debug material (RC:321-327). */
typedef struct RtSphere {
    float centre[3];
    float radius;
    RtMaterial material;
} RtSphere;

/* All per-dispatch uniforms of the kernel (RC:5-26,120-121; RCC:7-8), as set by
 * RCM:139-140,156,165-180,188-189,203. */
typedef struct RtParams {
    uint32_t abi_version;  /* = RT_ABI_VERSION                                  */
    uint32_t struct_size;  /* = sizeof(RtParams) — handshake                    */
    int32_t maxBounceCount;   /* RCM:168 */
    int32_t numRaysPerPixel;  /* RCM:169 */
    int32_t frame;            /* RCM:165,178: first frame after a reset is 1    */
    int32_t renderSeed;       /* RCM:179 */
    int32_t useSky;           /* RCM:166 */
    int32_t accumulate;       /* RCM:180 */
    float defocusStrength;    /* RCM:170 */
    float divergeStrength;    /* RCM:171 */
    float sunFocus;           /* RCM:173 */
    float sunIntensity;       /* RCM:174 */
    float sunColour[3];       /* RCM:175 (rgb of the Color)                     */
    float dirToSun[3];        /* RCM:176 */
    float viewParams[3];      /* RCM:188 (planeWidth, planeHeight, focusDist)   */
    float camLocalToWorld[16];/* RCM:189, column-major                          */
} RtParams;

/* Exact work counters of the frames rendered since the last rt_reset_counters.
 * They are the `stats` idea of RC:254,271 made observable; algorithmic bytes
 *  are computed from them.  `segments` is always exact; the
 * other fields are only filled by frames rendered while stats are enabled
 * (rt_enable_stats), otherwise they stay 0. */
typedef struct RtCounters {
    uint64_t segments;     /* CalculateRayCollision calls (RC:487)             */
    uint64_t innerSteps;   /* popped inner nodes (2 box tests each, RC:262-282) */
    uint64_t leafSteps;    /* popped leaf nodes (RC:248-261)                    */
    uint64_t triTests;     /* RayTriangle calls (RC:253)                        */
    uint64_t sphereTests;  /* RaySphere calls (extension)                       */
    uint64_t modelVisits;  /* model-loop iterations (RC:347)                    */
    uint64_t pixelFrames;  /* pixels written (W*H per frame)                    */
    double gpuMs;          /* device time between rt_timer_begin / rt_timer_end  */
} RtCounters;

/* Build statistics of rt_build_bvh — BVH.cs:518-576 */
typedef struct RtBvhStats {
    int32_t triangleCount, totalNodeCount, leafNodeCount;
    int32_t leafDepthMax, leafDepthMin, leafDepthSum;
    int32_t leafMaxTriCount, leafMinTriCount;
    int32_t quality;
    double timeMs;
} RtBvhStats;

typedef struct RtContext RtContext;

/* ---- lifetime: RCM:61-67 (OnEnable) / RCM:238-247 (OnDestroy -> Release) ---- */
/* Creates a context rendering on HIP device `device_id` (one process per GPU).
 * Fails with RT_ERR_NO_DEVICE if no MI355X-class HIP device is usable — there is
 * no CPU fallback. */
int rt_create(int device_id, RtContext** out);
void rt_destroy(RtContext* ctx);
const char* rt_last_error(const RtContext* ctx); /* ctx may be NULL: global msg */

/* Run launches on an existing hipStream_t (e.g. torch's current stream): every kernel of
 * this context is then enqueued on that stream, in call order.  NULL = the context's own
 * stream (the default); in that mode a frame may be launched as several kernels on internal
 * streams (RT_TWO_STREAMS=0 in the environment forbids it), and only the synchronising
 * calls — rt_synchronize, the rt_read_x / rt_display_x family, rt_get_counters — order host code
 * after them. */
int rt_set_stream(RtContext* ctx, void* hip_stream);

/* ---- render targets: RCM:126-141 InitTexturesAndBuffers -------------------- */
/* Global image resolution (uniform `Resolution`, RCM:139); (re)allocates the
 * library-owned FrameRender / AccumulatedRender for the rows this context owns
 * and zeroes the accumulator (CH:305-313 recreates textures on size change). */
int rt_resize(RtContext* ctx, int width, int height);

/* Multi-GPU image tiling (no reference counterpart — the reference is single
 * GPU): this context renders only strips s of `strip_rows` image rows with
 * s % part_count == part_index (cyclic), using GLOBAL pixel ids / Resolution so
 * every pixel is bit-identical to the single-GPU render.  The local buffers
 * hold the owned rows, packed in increasing global row order.  Default (1
 * part) = whole image.  strip_rows must be a multiple of 8.  Call before
 * rt_resize or it re-allocates. */
int rt_set_partition(RtContext* ctx, int strip_rows, int part_index, int part_count);
/* Number of image rows this context owns (after rt_resize). */
int rt_local_rows(const RtContext* ctx);
/* Global row index of local row `local_row`. */
int rt_local_to_global_row(const RtContext* ctx, int local_row);

/* Optional: render into caller-owned DEVICE buffers (≙ cs.SetTexture, RCM:135-137),
 * each local_rows*W*16 bytes, e.g. torch tensors to be gathered with RCCL.
 * Passing NULL for either returns to the library-owned buffer. */
int rt_bind_render_targets(RtContext* ctx, void* d_frame_render, void* d_accumulated);
/* Device pointers of the current targets (library-owned or bound).  Launches the frames rt_render_frame still holds
 * back; a host that reads the targets directly (after its own stream/device synchronise) instead of through rt_read_*
 * must call this or rt_flush/rt_synchronize first — frames are launched lazily (see rt_render_frame). */
int rt_get_render_targets(RtContext* ctx, void** d_frame_render, void** d_accumulated);

/* ---- scene: RCM:143-161 InitBVH (ComputeBuffer create + SetData) ----------- */
/* models/triangles/nodes are exactly the reference's three structured buffers;
 * spheres is the extension buffer (may be NULL/0).  Validates every index and
 * the depth of every BVH reachable from a model. */
int rt_upload_scene(RtContext* ctx,
                    const RtModel* models, int n_models,
                    const RtTriangle* triangles, int n_triangles,
                    const RtBVHNode* nodes, int n_nodes,
                    const RtSphere* spheres, int n_spheres);
/* The host half of rt_upload_scene without a device: the same validation and re-layout (node pairs, pre-differenced
 * triangles, root filters), nothing uploaded.  Returns the status rt_upload_scene would return for these buffers (message:
 * rt_last_error(NULL)); out_info (may be NULL) describes what would be uploaded.  For hosts that want to check a scene
 * before they own a GPU, and for the host-side tests. */
typedef struct RtSceneInfo {
    int32_t n_pairs;        /* sibling-pair records (64 bytes each) the traversal would fetch from: an UPPER BOUND — meshes converted by
                             * parallel workers get windows of the canonical array with slack between them (the uploaded pair space is
                             * compacted by every layout but `dense`) */
    int32_t max_height;     /* deepest BVH, in levels below a root: the traversal stack the scene needs */
    int32_t flat;           /* 1: every model's root is a leaf (no traversal stack at all) */
    int32_t n_filtered;     /* models behind the conservative root filter */
    float prepare_ms;       /* host time of the validation + re-layout */
} RtSceneInfo;
int rt_validate_scene(const RtModel* models, int n_models,
                      const RtTriangle* triangles, int n_triangles,
                      const RtBVHNode* nodes, int n_nodes,
                      const RtSphere* spheres, int n_spheres, RtSceneInfo* out_info);
/* RCM:192-204 UpdateModels: refresh matrices + materials (offsets must not change). */
int rt_update_models(RtContext* ctx, const RtModel* models, int n_models);
/* Refresh sphere centres/radii/materials (extension; count must not change). */
int rt_update_spheres(RtContext* ctx, const RtSphere* spheres, int n_spheres);

/* ---- uniforms: RCM:163-190 SetShaderParams + UpdateCameraParams ------------ */
int rt_set_params(RtContext* ctx, const RtParams* params);

/* ---- dispatch ------------------------------------------------------------- */
/* RCM:69-76 ResetAccumulatedRender → kernel ResetAccumulated (RCC:26-32); also
 * sets the context's frame counter to 1. */
int rt_reset_accumulation(RtContext* ctx);
/* RCM:84-95 RenderFrame → kernel RayTrace (RCC:10-24). Renders one frame with
 * uniform Frame = the context's frame counter (initialised from RtParams.frame
 * by rt_set_params), then increments the counter if accumulate (RCM:94).
 * Asynchronous: returns after enqueueing. */
int rt_render_frame(RtContext* ctx);
/* rt_render_frame only enqueues.  On the context's own stream, frames requested while earlier ones are still
 * executing are held back (at most 16) and leave as ONE fused launch — the form of rt_render_frames below, same
 * bits — as soon as 16 have gathered or at the next call that needs them: every call that changes what they
 * depend on (params, models, spheres, targets, size), rt_reset_accumulation, and every call that hands results to the
 * host (rt_synchronize, rt_read_x, rt_display_x, rt_get_counters, rt_timer_x).  An idle GPU is started at once.
 * rt_flush launches the held frames without waiting (for hosts that go away for a while after the last
 * rt_render_frame); RT_COALESCE=0 in the environment launches every frame at its call. */
int rt_flush(RtContext* ctx);
/* n consecutive frames Frame, Frame+1, ... : the same final FrameRender / AccumulatedRender
 * as n calls of rt_render_frame (same seeds, same per-pixel order of additions), but when
 * accumulating the frames are batched, up to 16 per launch, each pixel running its frames
 * back to back (RT_FUSE_FRAMES=0 in the environment restores one launch per frame). */
int rt_render_frames(RtContext* ctx, int n);
/* Wait for all enqueued work of this context. */
int rt_synchronize(RtContext* ctx);
/* Current frame counter (≙ numAccumulatedFrames, RCM:37). */
int rt_get_frame(const RtContext* ctx);

/* ---- readback (reference: none except ScreenCapture, RCM:106-111) --------- */
/* Copy the locally owned rows (local_rows*W*16 bytes) to host memory;
 * synchronises. `bytes` must be exactly that size. */
int rt_read_frame(RtContext* ctx, float* rgba, size_t bytes);
int rt_read_accumulated(RtContext* ctx, float* rgba, size_t bytes);

/* ---- display pass (RayTraceDisplay.cs:9-23 + Display.shader:42-47) ---------------- */
/* What the reference puts on screen: tex / Frame, where the caller passes
 * tex = accumulated (use_accumulated != 0, Frame = numAccumulatedFrames — the POST-increment
 * counter, so N accumulated frames are divided by N+1, the reference's off-by-one) or the last
 * frame (Frame = 1).  Host output, local rows, row 0 = bottom; synchronises. */
int rt_display(RtContext* ctx, int frame, int use_accumulated, float* rgba, size_t bytes);
/* The same followed by the linear->sRGB conversion and 8-bit quantisation of the back buffer
 * (Unity linear colour space); RGBA8, alpha 255; flip_y != 0 writes the top row first (PNG order). */
int rt_display_srgb8(RtContext* ctx, int frame, int use_accumulated, int flip_y, uint8_t* rgba8, size_t bytes);
/* Checkpoint/resume: restore the accumulation sum saved with rt_read_accumulated (the frame
 * counter and seed travel in RtParams).  local_rows*W*16 bytes. */
int rt_write_accumulated(RtContext* ctx, const float* rgba, size_t bytes);

/* ---- several GPUs from ONE host process (no reference counterpart) ------
 * For hosts that are not one-process-per-GPU launchers (the C# / C++ console hosts): n contexts, one
 * per entry of device_ids, context i owning the cyclic 8-row strips s % n == i of the image
 * (rt_set_partition(ctx_i, 8, i, n)).  The scene is replicated, every call below is forwarded to all
 * contexts (launches are asynchronous, so the devices run concurrently from one host thread), and the
 * only exchange is the gather at readback: rt_gather_accumulated / rt_gather_frame copy every
 * context's packed rows into their GLOBAL rows of a full H*W*16-byte host image (row 0 = bottom).
 * A device id may appear more than once (virtual shards on one GPU — how the 1-GPU tests cover this).
 * The image equals the single-context image bit for bit.  The one-process-per-GPU form of the same
 * tiling is rt_set_partition + rt_bind_render_targets + an RCCL gather (ray_tracing_amd/dist.py). */
typedef struct RtMulti RtMulti;
int rt_create_multi(const int* device_ids, int n_devices, RtMulti** out);
void rt_destroy_multi(RtMulti* m);
int rt_multi_count(const RtMulti* m);
/* The i-th context, for per-context calls (counters, timers, display, rt_last_error ...). */
RtContext* rt_multi_context(RtMulti* m, int i);
int rt_multi_resize(RtMulti* m, int width, int height);
/* Validated and re-laid out once on the host; context 0 is filled from the host, every other context copies context
 * 0's device arrays on its own stream (device to device, xGMI between different GPUs), all destinations at once. */
int rt_multi_upload_scene(RtMulti* m, const RtModel* models, int n_models, const RtTriangle* triangles, int n_triangles,
                          const RtBVHNode* nodes, int n_nodes, const RtSphere* spheres, int n_spheres);
int rt_multi_update_models(RtMulti* m, const RtModel* models, int n_models);
int rt_multi_update_spheres(RtMulti* m, const RtSphere* spheres, int n_spheres);
int rt_multi_set_params(RtMulti* m, const RtParams* params);
int rt_multi_reset_accumulation(RtMulti* m);
int rt_multi_render_frame(RtMulti* m);
int rt_multi_render_frames(RtMulti* m, int n);
/* Frames held back by any context are launched on ALL devices before the host waits for the first one. */
int rt_multi_synchronize(RtMulti* m);
/* bytes must be H*W*16.  Synchronises every context: held frames are launched everywhere, every device's tile is
 * copied to pinned host memory on its own stream (all devices concurrently) and scattered to its global rows. */
int rt_gather_accumulated(RtMulti* m, float* rgba, size_t bytes);
int rt_gather_frame(RtMulti* m, float* rgba, size_t bytes);
/* The same gather into DEVICE memory of context `root`'s GPU (H*W*16 bytes, e.g. a display or post-processing buffer):
 * every context's strips are copied device to device into their global rows on that context's own stream — xGMI between
 * different GPUs, all sources at once, no host memory in between.  Synchronises every context. */
int rt_gather_accumulated_to_device(RtMulti* m, int root, void* d_rgba, size_t bytes);
int rt_gather_frame_to_device(RtMulti* m, int root, void* d_rgba, size_t bytes);
/* Wall time of the last gather (copies + scatter, after the flush), milliseconds. */
double rt_multi_last_gather_ms(const RtMulti* m);
/* How copies between this multi-context's GPUs travel (peer access is checked and enabled by rt_create_multi): returns 2 = every
 * pair of distinct devices has direct peer access (xGMI / PCIe P2P), 1 = some, 0 = none (hipMemcpyPeerAsync then stages through
 * host memory: correct, slower), -1 = a single device; *pairs / *enabled (optional) receive the ordered-pair counts. */
int rt_multi_peer_access(const RtMulti* m, int* pairs, int* enabled);
/* Sum of the contexts' counters (gpuMs: the maximum). */
int rt_multi_get_counters(RtMulti* m, RtCounters* out);

/* ---- device timing ---------------------------------------------------------- */
/* HIP events recorded on the stream the kernels are launched on: the device
 * time between rt_timer_begin and rt_timer_end is added to RtCounters.gpuMs
 * (read by rt_get_counters, which synchronises). */
int rt_timer_begin(RtContext* ctx);
int rt_timer_end(RtContext* ctx);

/* ---- counters ------------------------------------------------------------- */
int rt_enable_stats(RtContext* ctx, int enabled); /* detailed counters on/off */
int rt_reset_counters(RtContext* ctx);
int rt_get_counters(RtContext* ctx, RtCounters* out); /* synchronises */

/* ---- host helpers so callers need not re-implement RCM/BVH host maths ------ */
/* BVH.cs:26-318: builds the reference-shaped flat BVH of one mesh.
 * verts/normals: n_verts*3 floats; indices: n_indices ints (3 per triangle).
 * out_nodes must hold >= 2*max(1,n_indices/3) nodes, out_tris n_indices/3
 * triangles. Pure host code (no device needed).
 * Returns RT_ERR_SCENE (nothing written to out_nodes, *out_n_nodes = 0) for input whose
 * reference-shaped tree would be malformed: coordinates so large that every split cost
 * overflows make BVH.cs peel off empty children (0-triangle leaves, which the shader reads as
 * inner nodes, RC:246) and more nodes than that capacity. */
int rt_build_bvh(const float* verts, const float* normals, int n_verts,
                 const int32_t* indices, int n_indices, int quality,
                 RtBVHNode* out_nodes, int* out_n_nodes,
                 RtTriangle* out_tris, RtBvhStats* out_stats);
/* The same builder on n_threads host threads (0 = auto: RT_BVH_THREADS or the core count,
 * at most 16).  Output is byte-identical to rt_build_bvh for every thread count: big nodes
 * are swept in ordered chunks reduced in order, subtrees are built privately and numbered
 * afterwards in the reference's allocation order.  rt_build_bvh itself uses the auto setting. */
int rt_build_bvh_mt(const float* verts, const float* normals, int n_verts,
                    const int32_t* indices, int n_indices, int quality, int n_threads,
                    RtBVHNode* out_nodes, int* out_n_nodes,
                    RtTriangle* out_tris, RtBvhStats* out_stats);
/* The same builder on the GPU `device_id` (host pointers in and out, like rt_build_bvh): level-synchronous,
 * byte-identical output — ordered chunk reductions for the sweeps, prefix sum + pointer jumping for the reference's
 * in-place partition, pre-order numbering for its node allocation order (ray-tracing_amd/csrc/rt_bvh_gpu.hip).
 * RT_ERR_NO_DEVICE without a HIP device. */
int rt_build_bvh_gpu(int device_id, const float* verts, const float* normals, int n_verts,
                     const int32_t* indices, int n_indices, int quality,
                     RtBVHNode* out_nodes, int* out_n_nodes,
                     RtTriangle* out_tris, RtBvhStats* out_stats);
/* The meshes of a scene in one call (what CreateAllMeshData does mesh by mesh, RCM:206-236): mesh k is built like rt_build_bvh_gpu
 * would build it and its nodes / triangles are written directly behind mesh k-1's — out_nodes (capacity: the sum of 2 * max(1,
 * triangles) over the meshes) and out_tris (the sum of the triangle counts) come out as the concatenated arrays the dispatcher uploads;
 * out_node_offset[k] / out_tri_offset[k] are mesh k's nodeOffset / triOffset (RC:79-80), out_n_nodes[k] its node count.
 * Two or more non-empty meshes are built as ONE forest (K roots, every level's kernels run once for the whole scene): twelve 82k-triangle
 * meshes cost what one 983k-triangle mesh costs.  A batch the forest refuses (bad index, degenerate mesh) is built mesh by mesh, so the
 * status and the meshes written before the offending one are the same either way. */
int rt_build_bvh_gpu_batch(int device_id, int n_meshes, const float* const* verts, const float* const* normals, const int* n_verts,
                           const int32_t* const* indices, const int* n_indices, int quality,
                           RtBVHNode* out_nodes, int* out_n_nodes, int* out_node_offset,
                           RtTriangle* out_tris, int* out_tri_offset, RtBvhStats* out_stats);
/* rt_build_bvh_gpu keeps its device scratch between calls (a scene build calls it once per mesh; at most 4 GiB is kept,
 * larger builds free it on return).  The scratch is ONE pool for the process: builds from several threads are serialised on
 * it, and this call frees it whichever thread built last. */
void rt_build_bvh_gpu_release(void);
/* RCM:183-190 UpdateCameraParams: fills viewParams from (fovDeg, aspect, focusDist). */
int rt_camera_view_params(float fov_deg, float aspect, float focus_distance, float out_view_params[3]);

/* ---- one process per GPU: the gather of the per-tile buffers over RCCL ----
 * The reference is single-GPU; this is the exchange step of the row-tiled multi-GPU form (BASELINE.json north_star: "a final
 * RCCL gather over xGMI of the per-tile accumulation buffer").  Every rank of the caller's communicator calls it with its
 * partitioned context (rt_set_partition(strip_rows, rank, world_size)): the packed tiles travel to `root` with ncclSend /
 * ncclRecv inside one group on the context's stream and are de-interleaved there into d_rgba (root only: H*W*16 bytes of
 * device memory on root's GPU, row 0 = bottom like every render target here; other ranks pass NULL / 0).
 * nccl_comm = a caller-owned ncclComm_t whose user rank / size are the context's partition index / count (checked).
 * use_accumulated 1 = AccumulatedRender, 0 = FrameRender.  Synchronous: returns when the image (root) / the send (others)
 * has completed.  RCCL (librccl.so) is loaded on first use; the library has no link-time dependency on it — without RCCL
 * the call fails with RT_ERR_STATE and everything else works.
 * COLLECTIVE — HAZARD: every rank of the communicator must make this call, and a rank that returns early leaves its peers blocked
 * inside ncclSend / ncclRecv.  What can differ per rank is therefore kept out of the way of the exchange: a root whose d_rgba / bytes
 * are wrong still RECEIVES every tile (into scratch) and reports RT_ERR_INVALID_ARG after the group has completed, so the senders
 * return normally.  Errors that every rank sees alike (null communicator, communicator != partition, root out of range, no image
 * size) return before the exchange on all of them.  Two failures remain one-sided and are the caller's to recover from with
 * ncclCommAbort on the peers: pending frames that fail to launch on one rank, and a root that cannot allocate its H*W*16-byte
 * staging area. */
int rt_gather_rccl(RtContext* ctx, void* nccl_comm, int root, int use_accumulated, void* d_rgba, size_t bytes);

/* ---- test hooks: the kernel's device functions on caller-supplied inputs ---- */
/* CalculateRayCollision (RC:335-374) for n world rays (origins/dirs: n*3 floats,
 * host memory). out10 per ray: didHit, isBackface, dst, normal.xyz, pos.xyz,
 * material.flag. Synchronous. */
int rt_debug_intersect(RtContext* ctx, const float* origins, const float* dirs, int n, float* out10);
/* Device evaluation of the include/rt_math.h primitives over host arrays.
 * op: 0 log, 1 exp, 2 sin, 3 cos, 4 sqrt, 5 pow(x,y), 6 x/y, 7 smoothstep(0,y,x). */
int rt_debug_math_eval(RtContext* ctx, int op, const float* x, const float* y, float* out, int n);

/* The device-memory layout rt_upload_scene would produce for these buffers, on the host (no device needed) — for the
 * host-side layout tests and tools/layout_sim: every record is named by the 16-byte unit it starts at (inner code = unit of
 * a 64-byte pair record in pair_space; leaf code = bit 31 | count << 24 | first unit of the run of 48-byte triangle
 * records relative to tri_base[model], count 0 = index into big_leaves {unit, count}; norm_space holds 12 bytes per unit of
 * the triangle space).  `layout` = an RT_LAYOUT string or NULL (environment / default).  arena = 1: the triangles live in
 * pair_space (tri_space is NULL).  Free with rt_debug_layout_free. */
typedef struct RtLayoutDump {
    unsigned char* pair_space; size_t pair_bytes;
    unsigned char* tri_space;  size_t tri_bytes;
    unsigned char* norm_space; size_t norm_bytes;
    uint32_t* big_leaves;      size_t n_big_leaves; /* pairs of (unit, count) */
    uint32_t* root_codes;      /* per model */
    int32_t* tri_base;         /* per model, units */
    int32_t n_models;
    int32_t arena;
    char used[64];             /* the layout that was applied (an irregular scene gets "dense") */
} RtLayoutDump;
int rt_debug_layout(const RtModel* models, int n_models, const RtTriangle* triangles, int n_triangles,
                    const RtBVHNode* nodes, int n_nodes, const char* layout, RtLayoutDump* out);
void rt_debug_layout_free(RtLayoutDump* dump);

/* Wave-level divergence profile of the frames rendered with stats enabled since the
 * last rt_reset_counters: out[2p] = times a wave executed phase p, out[2p+1] = lanes
 * active in it (p: 0 loop, 1 camera ray, 2 spheres, 3 traverse call, 4 model setup,
 * 5 inner step, 6 triangle test, 7 shade hit, 8 sky, 9 sphere roots, 10 glass branch,
 * 11 pixel refill). n must be >= 24; if n >= 25, out[24] = number of times the conservative
 * world-space root filter rejected a model the exact root step would have entered (must be 0);
 * if n >= 28 (round 6), in lane-steps of the inner phase: out[25] = served by the LDS top-of-tree cache,
 * out[26] = taken while >= 48 lanes of the wave stood on one node, out[27] = while >= 3/4 of >= 16 active lanes did;
 * scenes without trees (every model a single leaf): out[25] = pixel chains handed from one wave to another through the
 * workgroup's LDS chain pool (deposits; rt_kernels.h, pool_exchange), out[26] = out[27] = 0. */
int rt_debug_phase_profile(RtContext* ctx, uint64_t* out, int n);

/* Frames the context would put into one fused launch right now (16 ... 64: a budget that follows the measured frame time, see
 * rt_context.hip RT_FUSE_MIN; scheduling only, results never depend on it). */
int rt_debug_fused_frames_cap(const RtContext* ctx);

/* Library identification: returns "raytrace_hip gfx950 abi=<n>" */
const char* rt_version(void);

#ifdef __cplusplus
} /* extern "C" */

static_assert(sizeof(RtMaterial) == 88, "RtMaterial must be 88 bytes");
static_assert(sizeof(RtModel) == 224, "RtModel must be 224 bytes");
static_assert(sizeof(RtTriangle) == 72, "RtTriangle must be 72 bytes");
static_assert(sizeof(RtBVHNode) == 32, "RtBVHNode must be 32 bytes");
static_assert(sizeof(RtSphere) == 104, "RtSphere must be 104 bytes");
#endif

#endif /* RT_ABI_H */
