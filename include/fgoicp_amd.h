/*
 * fgoicp_amd.h — C ABI of the MI355X-native Go-ICP hot path (libfgoicp_amd.so).
 *
 * The reference (solemnwind/fast-go-icp) has no FFI layer: its boundary is the C++ class
 * interface in namespace icp.  Every entry point below names the reference
 * interface it replaces (paths relative to the reference checkout).  Plain pointers and sizes
 * only; every function returns an fgoicp_status (0 = OK) and never throws across the ABI.
 *
 * Conventions
 *   - points:    float xyz triples, tightly packed (glm::vec3 / icp::Point3D, common.hpp:130).
 *   - matrices:  9 floats in glm::mat3 memory order = COLUMN-major, m[col*3 + row]
 *                (common.hpp:50-54).
 *   - all pointers are HOST pointers unless a parameter says "device".
 *   - a context is bound to one HIP device and one stream; it is not thread-safe.
 *   - there is NO CPU fallback: without a usable HIP device every create call fails with
 *     FGOICP_ERR_NO_DEVICE.
 */
#ifndef FGOICP_AMD_H
#define FGOICP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum fgoicp_status {
    FGOICP_OK = 0,
    FGOICP_ERR_INVALID_ARG = 1,
    FGOICP_ERR_NO_DEVICE = 2,   /* no HIP device / HIP runtime error at init            */
    FGOICP_ERR_HIP = 3,         /* a HIP call failed (see fgoicp_last_error)             */
    FGOICP_ERR_OOM = 4,
    FGOICP_ERR_TOO_LARGE = 5,   /* batch larger than the context was sized for          */
    FGOICP_ERR_EXCHANGE = 6     /* the multi-GPU exchange callback reported a failure   */
} fgoicp_status;

/* Thread-local description of the last failure on the calling thread ("" if none). */
const char* fgoicp_last_error(void);
const char* fgoicp_version(void);
/* 1 = this is the development build (-DFGOICP_DEV_KNOBS, libfgoicp_amd_dev.so): it reads the FGOICP_* tuning / A-B variables of
 * NOTES.md from the environment and carries the kernel variants that were measured and rejected.  0 = the shipped build: it reads
 * FGOICP_HOST_THREADS / FGOICP_HOST_SPIN only and no stray variable can change its code path. */
int fgoicp_dev_knobs(void);
/* The ABI revision of this header.  2 (round 4): fgoicp_exchange and fgoicp_ctx_info start with `struct_size` (members may be appended
 * from now on without breaking callers built against this revision); fgoicp_solver_set_log, fgoicp_rccl_create_ex added.  A caller
 * built against revision 1 (no struct_size) must be rebuilt: INTEGRATION.md "ABI revisions". */
#define FGOICP_ABI_VERSION 2
int fgoicp_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Operator level: icp::Registration + icp::NearestNeighborLUT + icp::IterativeClosestPoint3D
 * ------------------------------------------------------------------------------------------ */
typedef struct fgoicp_ctx fgoicp_ctx;

enum {
    FGOICP_FLAG_NO_WEIGHT_QUANT = 1u << 0, /* trilinear weights in full fp32 instead of CUDA's 1.8 fixed point */
    FGOICP_FLAG_NO_MORTON       = 1u << 1, /* keep the source cloud in caller order on the device             */
    FGOICP_FLAG_PROFILE         = 1u << 2, /* bracket every bounds kernel with HIP events (fgoicp_ctx_profile) */
    FGOICP_FLAG_BRUTE_FORCE_NN  = 1u << 3, /* O(n*m) brute-force kernels for LUT build / SSE / ICP instead of the exact BVH */
    FGOICP_FLAG_CURVE_ORDER     = 1u << 4  /* source cloud along the Hilbert curve instead of in k-d order on the device: for sources with outliers
                                              spread through the volume (trimmed runs; fgoicp_solver_create sets it when trim_fraction > 0).  Locality only:
                                              results do not depend on the order */
};

/*
 * Replaces icp::Registration::Registration(pct, pcs, target_bounds, lut_resolution)
 * (fgoicp/registration.hpp:68-80) together with icp::NearestNeighborLUT::NearestNeighborLUT /
 * build (fgoicp/registration.cu:180-207, 258-318): uploads both clouds, builds the
 * nearest-squared-distance LUT on the device.  bounds6 = {minx,maxx,miny,maxy,minz,maxz} of the
 * target (std::array<std::pair<float,float>,3>).  Clouds are copied; the caller keeps ownership.
 */
int fgoicp_ctx_create(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, const float* bounds6,
                      float lut_resolution, int device, unsigned flags, fgoicp_ctx** out);
/* Replaces ~Registration / ~NearestNeighborLUT (registration.hpp:82-87, registration.cu:209-248). */
void fgoicp_ctx_destroy(fgoicp_ctx* ctx);

/* LUT geometry / contents in the reference's layout, index (z*dy + y)*dx + x
 * (registration.cu:186-188, :276).  Used by tests and by INTEGRATION.md's façade. */
int fgoicp_lut_dims(const fgoicp_ctx* ctx, int* dims3);
int fgoicp_lut_read(fgoicp_ctx* ctx, float* out, size_t capacity_floats);
/* What the context derived from the clouds' statistics for its own scaling decisions (the reference's open item "compute point
 * clouds' stats, normalize (For scalability)", TODO.md:7): LUT size in every layout it keeps, source points per voxel of the LUT's
 * faces (the density that picks the packed layout and the points per work item), work items per evaluation. */
typedef struct fgoicp_ctx_info {
    size_t struct_size;          /* IN: sizeof(fgoicp_ctx_info) as the CALLER was compiled (ABI 2).  fgoicp_ctx_get_info writes no byte beyond it, so a caller
                                    built against an older, shorter struct is not overrun when members are appended; 0 is refused */
    int lut_dims[3];
    int lut_layout;              /* next to the plain fp32 LUT: 1 = z-pair copy (8 B per node), 2 = yz-quad copy (16 B per node), 4 = apron-bricked yz-quad copy (21.3 B per node) */
    uint64_t lut_nodes;
    uint64_t lut_bytes;          /* plain LUT (padded) + packed copy, device memory */
    double source_points_per_face_voxel;
    int points_per_item;         /* 256 .. 2048 */
    int items_per_evaluation;    /* ceil(ns / points_per_item) */
    int max_subcubes_per_window;
    int source_order;            /* order of the source cloud on the device: 0 = caller / Z-order, 1 = Hilbert curve, 2 = k-d cells of 64 points, 3 = density split */
    int tree_order;              /* leaves of the target tree: 1 = k-d cells of 32 points, 0 = runs of the space-filling curve */
    int chunks_per_item_with_thresholds; /* fgoicp_bounds_submit_cut: a work item of such a submission spans this many chunks of points_per_item points (1 or 2).  Appended within revision 2: a caller built with the shorter struct is served as before (struct_size) */
} fgoicp_ctx_info;
int fgoicp_ctx_get_info(const fgoicp_ctx* ctx, fgoicp_ctx_info* out);
/* n single nodes of the LUT: out[i] = node (x, y, z) = xyz[3i..3i+2] (the value buildLUTKernel, registration.cu:258-278, stores at
 * index (z*dy + y)*dx + x).  For LUTs too large to read back whole (test/bunny.toml: 923 x 906 x 711 nodes). */
int fgoicp_lut_nodes(fgoicp_ctx* ctx, const int* xyz, size_t n, float* out);
/* Device evaluation of NearestNeighborLUT::search (registration.cu:320-328; CUDA tex3D linear
 * filtering restated in software) at n query points. */
int fgoicp_lut_search(fgoicp_ctx* ctx, const float* q_xyz, size_t n, float* out);

/*
 * Replaces Registration::compute_sse_error(RotNode&, std::vector<TransNode>&, bool, StreamPool&)
 * (registration.hpp:97, registration.cu:88-152; kernel kernComputeBounds :27-60).
 * tnodes4 = B x {t.x, t.y, t.z, span}.  Outputs: lb_out[B], ub_out[B] (the reference returns
 * {lower, upper}, registration.cu:151).  B may exceed the reference's 32.
 */
int fgoicp_bounds_batch(fgoicp_ctx* ctx, const float* R9, float rot_span, const float* tnodes4, int B, int fix_rot,
                        float* lb_out, float* ub_out);
/*
 * The same operator for G rotation nodes in one submission (one host<->device round trip):
 * group g uses R9[g*9..], rot_span[g], fix_rot[g] and the translation nodes
 * tnodes4[offsets[g]*4 .. offsets[g+1]*4).  lb_out/ub_out are indexed like tnodes.
 */
int fgoicp_bounds_multi(fgoicp_ctx* ctx, int G, const float* R9, const float* rot_span, const int* fix_rot,
                        const int* offsets, const float* tnodes4, float* lb_out, float* ub_out);

/*
 * Asynchronous form of fgoicp_bounds_multi on one of two slots (0, 1), each with its own buffers (the
 * kernels of both queue on the context's stream and run back to back): submit returns once the work is queued, collect waits for it and returns the bounds in
 * submission order.  Lets the host prepare the next submission of one slot while the device works on
 * the other.  A slot must be collected before it is submitted again.
 */
int fgoicp_bounds_submit(fgoicp_ctx* ctx, int slot, int G, const float* R9, const float* rot_span, const int* fix_rot,
                         const int* offsets, const float* tnodes4);
/*
 * The same with a hint: twin[i] = j (and twin[j] = i) says that subcubes i and j of this submission are the SAME translation
 * node under the SAME rotation, once in a fix_rot group and once in a non-fix_rot group — what the UB and the LB inner
 * branch-and-bound of one rotation cube (fgoicp.cpp:69, :90) produce while they walk the top of the same translation tree.
 * Such a pair is evaluated with one LUT lookup per point and both variants of the bound formulae (registration.cu:39-58);
 * the four sums are bit-identical to two separate evaluations.  twin[i] = -1 (or twin = NULL): no twin.  The hint is
 * validated (rotation, span, node and fix_rot are compared), a wrong hint only costs the saving.
 */
int fgoicp_bounds_submit_twins(fgoicp_ctx* ctx, int slot, int G, const float* R9, const float* rot_span, const int* fix_rot,
                               const int* offsets, const float* tnodes4, const int* twin);
/*
 * The same with what the caller does NOT need to know: cut_above[g] = T (one per group; +inf, or cut_above = NULL: everything) says
 * that for a subcube of group g whose lower bound is >= T the exact bounds do not matter — the inner branch-and-bound drops such a node
 * whatever they are (fgoicp.cpp:151, with T = its running best error) and its result takes part in comparisons only (:74, :92;
 * csrc/host/driver.hpp InnerTask::cut_above).  Such a row comes back as lb = ub = T.  All terms of the lower-bound sum are >= 0 and the
 * evaluation of a subcube is spread over many work items (chunks of source points), so the kernel stops evaluating a subcube once the
 * sums of its finished items have reached T: on a certify run two thirds of all point evaluations belong to such subcubes
 * (profiles/r04_early_exit_*).  Rows below their threshold are bit-identical to fgoicp_bounds_submit_twins; the answer for a row at or
 * above it is {T, T} whether the kernel got to cut it short or not (deterministic).  Trimmed contexts ignore cut_above.
 */
int fgoicp_bounds_submit_cut(fgoicp_ctx* ctx, int slot, int G, const float* R9, const float* rot_span, const int* fix_rot,
                             const int* offsets, const float* tnodes4, const int* twin, const float* cut_above);
int fgoicp_bounds_collect(fgoicp_ctx* ctx, int slot, float* lb_out, float* ub_out);
/* Work items (subcube x chunk of source points) of the submissions that carried thresholds since the last reset, and how many of them
 * the early exit did not evaluate (measurement: bench.py prices the kernel on the items it evaluated).  Call between submissions. */
int fgoicp_ctx_cut_stats(fgoicp_ctx* ctx, uint64_t* items_offered, uint64_t* items_cut, int reset);

/* Replaces float Registration::compute_sse_error(glm::mat3 R, glm::vec3 t)
 * (registration.hpp:96, registration.cu:62-86; kernels :14-25, :154-174): exact nearest
 * neighbour SSE of R*src + t against the target. */
int fgoicp_sse(fgoicp_ctx* ctx, const float* R9, const float* t3, float* sse_out);

/* Replaces IterativeClosestPoint3D(reg, pct, pcs, max_iter, thr, R, t) + run()
 * (fgoicp/icp3d.hpp:30-35, icp3d.cu:55-108).  Returns the reference's Result_t {sse, R, t}
 * plus the number of loop iterations executed. */
int fgoicp_icp(fgoicp_ctx* ctx, const float* R0_9, const float* t0_3, size_t max_iter, float conv_thr, float* sse_out,
               float* R_out9, float* t_out3, int* iters_out);
/* n IterativeClosestPoint3D objects (fgoicp.cpp:76-77 creates and runs one per promising rotation cube) run at once.  Run i starts from R0s_9[9i..], t0s_3[3i..] and is exactly the run
 * fgoicp_icp would do; runs share the device through per-run scratch and streams (an ICP iteration is a chain of small
 * kernels: below ~100k points one run fills a fraction of the device). */
int fgoicp_icp_batch(fgoicp_ctx* ctx, int n, const float* R0s_9, const float* t0s_3, size_t max_iter, float conv_thr, float* sse_out,
                     float* R_out9s, float* t_out3s, int* iters_out);
/* One IterativeClosestPoint3D::procrustes() step (icp3d.cu:140-172) on an explicit working cloud
 * (ns x xyz, caller order).  Test hook: optional outputs may be NULL.  In trimmed mode corr_idx is 0x7fffffff for points
 * that provably lie outside the inlier set (they take no part in the step and their exact neighbour is not searched). */
int fgoicp_procrustes(fgoicp_ctx* ctx, const float* working_xyz, float* R_out9, float* t_out3, float* centroids6,
                      float* ABt9, int* corr_idx);

/*
 * EXTENSION — trimmed Go-ICP.  The reference parses `params.trim` (src/utilities.hpp:94) but never uses it, so
 * there is no reference behaviour; this follows Yang et al.'s Go-ICP: with k inliers every sum over source
 * points (bounds, SSE, the Procrustes means) runs over the k smallest per-point terms.  k = 0 or k >= ns: off.
 */
int fgoicp_ctx_set_inliers(fgoicp_ctx* ctx, size_t k);

/*
 * Trimmed mode only (diagnostic): the per-point quantity behind one subcube's bounds, in caller order —
 * e_i = max(d_i, 0) with d_i the value of `distance` at registration.cu:48-52 (after the rotation-uncertainty term when
 * fix_rot == 0); the reference's per-point outputs are ub_i = e_i^2 (:54) and lb_i = max(e_i - sqrt3*span, 0)^2 (:57-58).
 */
int fgoicp_bounds_point_distances(fgoicp_ctx* ctx, const float* R9, float rot_span, const float* tnode4, int fix_rot, float* e_out);

/* Sorted ticks so far and how many of them had to be repeated because the on-device check found that the locality sort had not
 * produced a permutation of the work items (then the context switches to device-scope atomics for good). */
int fgoicp_ctx_sort_fallbacks(const fgoicp_ctx* ctx, uint64_t* sorted_ticks, uint64_t* fallbacks);
/* TEST HOOK, not part of the drop-in surface: spoils the nth_tick-th sorted tick from now (0 = off) so that the permutation check
 * above has something to find.  (Round 2 read this from the environment of the shipped library) */
int fgoicp_ctx_test_sort_fault(fgoicp_ctx* ctx, int nth_tick);

/* Accumulated HIP-event timing of the bounds kernel since the last reset (FGOICP_FLAG_PROFILE):
 * kernel_ms = sum of launch durations, launches = kernel launches, subcubes = (rot, trans) pairs. */
int fgoicp_ctx_profile(fgoicp_ctx* ctx, double* kernel_ms, uint64_t* launches, uint64_t* subcubes, int reset);
/* Evaluations behind those subcubes since the last reset: a twin pair (fgoicp_bounds_submit_twins) is two subcubes and one
 * evaluation.  Read it before the fgoicp_ctx_profile call that resets. */
int fgoicp_ctx_profile_evaluations(fgoicp_ctx* ctx, uint64_t* evaluations);
/* Trimmed mode: accumulated duration of the selection kernel (one launch per window, next to the bounds kernel) since the last
 * reset by fgoicp_ctx_profile — read it before that call. */
int fgoicp_ctx_profile_select_ms(fgoicp_ctx* ctx, double* select_ms);
/* Trimmed mode (EXTENSION): how the per-row selections of the bounds went since the last reset — out3 = {rows selected, rows whose
 * sampled bracket failed its exact check and that were done again in two passes, bracket members gathered}.  No submission may
 * be in flight. */
int fgoicp_ctx_trim_stats(fgoicp_ctx* ctx, uint64_t* out3, int reset);
/* Multi-rank runs: from how many source points a cooperative refinement (fgoicp_exchange.allgather_device) splits its exact scans over the
 * ranks instead of running replicated on every rank.  Defaults: untrimmed never ((size_t)-1: measured no faster at 437k points on 8 ranks),
 * trimmed contexts from 262 144 points (their iterations are long: 8-rank replay 1.84x -> 2.69x).  A deployment choice (it trades an
 * in-place device all-gather of 4 B per source point, twice per ICP iteration, against the replicated scan), hence an option, not a knob. */
int fgoicp_ctx_set_coop_split(fgoicp_ctx* ctx, size_t min_points_untrimmed, size_t min_points_trimmed);
/* Turns the HIP-event bracketing on or off at run time (events are created on first use). */
int fgoicp_ctx_set_profile(fgoicp_ctx* ctx, int enabled);
size_t fgoicp_ctx_ns(const fgoicp_ctx* ctx);
size_t fgoicp_ctx_nt(const fgoicp_ctx* ctx);

/* ------------------------------------------------------------------------------------------
 * Driver level: icp::FastGoICP (fgoicp/fgoicp.hpp:13-43, fgoicp/fgoicp.cpp:10-287)
 * ------------------------------------------------------------------------------------------ */
typedef struct fgoicp_solver fgoicp_solver;

typedef enum fgoicp_schedule {
    FGOICP_SCHEDULE_SERIAL = 0, /* the reference's exact exploration order (fgoicp.cpp:32-174)       */
    FGOICP_SCHEDULE_ROUND  = 1  /* expansion rounds: pop K cubes, evaluate all children concurrently  */
} fgoicp_schedule;

/*
 * Exchange hook for the sharded outer BnB (one process per GPU).  Called once per expansion
 * round on every rank, from the thread that called fgoicp_solver_run:
 *   allreduce_min(buf, n, user)            — in-place element-wise MIN over ranks (RCCL all-reduce)
 *   allgather(send, recv, n_per_rank, user)— recv[r*n .. (r+1)*n) = rank r's send[0..n)
 * Both return 0 on success.
 */
typedef struct fgoicp_exchange {
    size_t struct_size;  /* sizeof(fgoicp_exchange) as the caller was compiled (ABI 2): members beyond it are taken as NULL / 0, so a struct from an older
                            header (without allgather_device, say) never hands the library a garbage pointer; 0 is refused */
    int rank;
    int world_size;
    int (*allreduce_min)(float* buf, size_t n, void* user);
    int (*allgather)(const float* send, float* recv, size_t n_per_rank, void* user);
    void* user;
    /* Optional (NULL = off).  In-place all-gather on DEVICE memory of the calling rank's context: on return
     * device_buf[r * bytes_per_rank .. (r + 1) * bytes_per_rank) holds rank r's chunk, for every r (RCCL: ncclAllGather with
     * sendbuff = recvbuff + rank * count).  With it a round's refinements become COOPERATIVE: the child bounds are exchanged first,
     * every rank applies the trigger rule of fgoicp.cpp:74-88 to ALL children in the single-GPU order, and each triggered
     * IterativeClosestPoint3D::run() is executed by all ranks together — every rank scans 1/world of the source for the two exact
     * nearest-neighbour passes of an iteration, the per-query results (4 B each) are all-gathered through this hook, the sums and the
     * SVD are replicated.  Results are the single-GPU bits on every rank.  Called twice per ICP iteration, from the thread that
     * called fgoicp_solver_run; the context's streams are idle during the call. */
    int (*allgather_device)(void* device_buf, size_t bytes_per_rank, void* user);
} fgoicp_exchange;

typedef struct fgoicp_solver_opts {
    int schedule;        /* fgoicp_schedule                                            */
    int round_width;     /* ROUND: rotation cubes popped per expansion round (K >= 1); 0 = adaptive: 32 per rank,
                            doubled after every round that does not improve the incumbent, reset when one does */
    unsigned ctx_flags;  /* FGOICP_FLAG_*                                              */
    int device;          /* HIP device ordinal                                         */
    float trim_fraction; /* EXTENSION: fraction of source points treated as outliers (0 = the reference's behaviour) */
} fgoicp_solver_opts;

typedef struct fgoicp_run_stats {
    uint64_t trans_cubes;   /* (rot, trans) subcubes evaluated = `count` of fgoicp.cpp:108,132 */
    uint64_t bounds_calls;  /* bounds-operator submissions                                      */
    uint64_t rot_cubes;     /* rotation cubes that went through branch_and_bound_R3             */
    uint64_t icp_runs;
    uint64_t icp_iters;
    uint64_t inner_bnb;
    uint64_t rounds;        /* expansion rounds (ROUND) / popped rotation nodes (SERIAL)        */
    double   seconds_total; /* wall-clock of run()                                              */
    double   seconds_bnb;   /* of which: outer BnB phase                                        */
    double   seconds_icp;   /* of which: inside ICP                                             */
    double   initial_icp_sse; /* error of the initial ICP from identity (fgoicp.cpp:12-17)        */
} fgoicp_run_stats;

/* Replaces FastGoICP::FastGoICP(pct, pcs, lut_resolution, mse_threshold) (fgoicp.hpp:13-25):
 * centre, scale, bounds (fgoicp.cpp:176-287), then the Registration constructor. */
int fgoicp_solver_create(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, float lut_resolution,
                         float mse_threshold, const fgoicp_solver_opts* opts, fgoicp_solver** out);
void fgoicp_solver_destroy(fgoicp_solver* s);
/* Installs the multi-GPU exchange (NULL = single process). */
int fgoicp_solver_set_exchange(fgoicp_solver* s, const fgoicp_exchange* ex);
/* The log lines FastGoICP::run() and branch_and_bound_SO3 emit while they run (fgoicp.cpp:15-17, :85-87), as a callback on the thread
 * that called fgoicp_solver_run:
 *   FGOICP_LOG_INITIAL_ICP  once, after the initial ICP: its (sse, R, t) as it returned them (t in the solver's scaled frame, as the
 *                           reference prints it);
 *   FGOICP_LOG_NEW_BEST     after EVERY refinement the search triggers (improved or not, as the reference): the incumbent's error and
 *                           rotation and its translation RESTORED to the callers' frame (restore_translation, fgoicp.hpp:87-90).
 * Under FGOICP_SCHEDULE_SERIAL the events come in the reference's order.  cb = NULL removes the hook.  icp::FastGoICP (include/fgoicp/
 * fgoicp.hpp) installs one that prints the reference's lines through icp::Logger. */
enum { FGOICP_LOG_INITIAL_ICP = 0, FGOICP_LOG_NEW_BEST = 1 };
typedef void (*fgoicp_log_fn)(int event, float sse, const float* R9, const float* t3, void* user);
int fgoicp_solver_set_log(fgoicp_solver* s, fgoicp_log_fn cb, void* user);
/* on = 1 (default): the inner branch-and-bounds hand their thresholds to the bounds operator (fgoicp_bounds_submit_cut); 0: every
 * subcube is evaluated in full, as the reference does.  Same trajectory, counters and result either way. */
int fgoicp_solver_set_early_exit(fgoicp_solver* s, int on);
/* Replaces FastGoICP::run() (fgoicp.cpp:10-30): returns R and the restored translation
 * (fgoicp.hpp:87-90). */
int fgoicp_solver_run(fgoicp_solver* s, float* R_out9, float* t_out3);
/* Replaces get_best_error / get_best_transform / get_last_transform (fgoicp.hpp:33-43);
 * translations here are in the solver's scaled frame, as in the reference. */
int fgoicp_solver_best_error(const fgoicp_solver* s, float* sse_out);
int fgoicp_solver_best_transform(const fgoicp_solver* s, float* R9, float* t3);
int fgoicp_solver_last_transform(const fgoicp_solver* s, float* R9, float* t3);
int fgoicp_solver_stats(const fgoicp_solver* s, fgoicp_run_stats* out);
/* Pre-processing results (tests): offs6 = {offset_pcs, offset_pct}, bounds6 as in ctx_create. */
int fgoicp_solver_preproc(const fgoicp_solver* s, float* offs6, float* scale, float* bounds6);
/* Statistics of a raw cloud, host side, no device needed (TODO.md:7 of the reference: "compute point clouds' stats"): what the
 * pre-processing (fgoicp.cpp:176-287) derives its centring and scaling from, plus the spread.  Sums in double. */
typedef struct fgoicp_cloud_stats_t {
    uint64_t n;
    float centroid[3];
    float min[3], max[3];        /* axis-aligned bounding box */
    float max_abs_centred;       /* max_i max(|x_i - c_x|, |y_i - c_y|, |z_i - c_z|): 1 / this is the scale of fgoicp.cpp:205-220 when taken over the source */
    float rms_radius;            /* sqrt(mean |p_i - c|^2) */
} fgoicp_cloud_stats_t;
int fgoicp_cloud_stats(const float* xyz, size_t n, fgoicp_cloud_stats_t* out);
/* The operator context the solver drives (borrowed; valid until solver_destroy). */
fgoicp_ctx* fgoicp_solver_ctx(fgoicp_solver* s);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU inside the library (no reference counterpart: the reference is single-GPU).
 * The rotation cubes of every expansion round are dealt round-robin to the ranks; per round ONE all-reduce(MIN) of the
 * best error and ONE small all-gather keep every rank's queue and incumbent identical.
 * ------------------------------------------------------------------------------------------ */
typedef struct fgoicp_rccl fgoicp_rccl;
/* RCCL transport of the exchange, one communicator per rank (one process per GPU, or one thread per GPU): rank 0 draws the
 * 128-byte id (ncclGetUniqueId) and hands it to the others by whatever channel the launcher has; every rank then calls
 * fgoicp_rccl_create (ncclCommInitRank — blocks until all ranks have joined) and installs fgoicp_rccl_exchange's struct with
 * fgoicp_solver_set_exchange.  The collectives run on device buffers over xGMI, on a stream of their own. */
int fgoicp_rccl_unique_id(unsigned char* id128);
/* Which librccl the transport runs on (loads it on first use): the instance already mapped into the process — PyTorch-ROCm brings its own
 * librccl.so + librocm_smi64.so, and a second copy next to it makes the process abort in its exit handlers (two librocm_smi64 destroying
 * the same globals) — or the system's librccl.so.1, loaded RTLD_LOCAL | RTLD_DEEPBIND.  Returns a description (or the load error). */
const char* fgoicp_rccl_library(void);
int fgoicp_rccl_create(int rank, int world_size, const unsigned char* id128, int device, fgoicp_rccl** out);
/* The same with the kind of communicator chosen: nonblocking != 0 creates it with ncclConfig_t.blocking = 0 (what fgoicp_multi_create
 * does for its rank threads, so that a rank can abandon the set-up when a peer fails); every collective on such a communicator may
 * return ncclInProgress and is polled (ncclCommGetAsyncError) until it is on the stream before anything else is enqueued. */
int fgoicp_rccl_create_ex(int rank, int world_size, const unsigned char* id128, int device, int nonblocking, fgoicp_rccl** out);
/* TEST HOOK, not part of the drop-in surface: the next n collectives of x report ncclInProgress once before their real status (the
 * polling path on a box where RCCL answers at once); settled_out (optional) = polls of ncclCommGetAsyncError made so far. */
int fgoicp_rccl_test_inprogress(fgoicp_rccl* x, int n, uint64_t* settled_out);
int fgoicp_rccl_exchange(fgoicp_rccl* x, fgoicp_exchange* out);   /* `out` borrows x: keep x alive while a solver uses it */
int fgoicp_rccl_calls(const fgoicp_rccl* x, uint64_t* collectives);
/* Ranks the communicator itself reports (ncclCommCount) — printed by bench.py so that a scaling run shows RCCL joined N ranks. */
int fgoicp_rccl_comm_count(fgoicp_rccl* x, int* count);
/* A PEER rank has failed: the collective in flight on x and every later one must end instead of waiting for it for ever.  Any
 * thread, once or more — it only raises a flag; the thread that runs x's collectives polls it while it waits and aborts the
 * communicator itself (ncclCommAbort frees it: it must not be called under a running collective's feet).  x stays valid until
 * fgoicp_rccl_destroy but cannot be used for another run. */
int fgoicp_rccl_abort(fgoicp_rccl* x);
void fgoicp_rccl_destroy(fgoicp_rccl* x);

/* One process, one host thread + one solver per device — what `fast-go-icp --gpus N` runs.  devices[r] = HIP ordinal of rank r.
 * FGOICP_TRANSPORT_RCCL needs distinct devices; FGOICP_TRANSPORT_IN_PROCESS (a shared-memory rendezvous of the rank threads)
 * takes any list, e.g. {0, 0, 0, 0}: four ranks rehearsed on one GPU.  opts->schedule selects what is sharded (opts == NULL:
 * ROUND, adaptive width): FGOICP_SCHEDULE_ROUND deals the children of an expansion round over the ranks (fastest; an epsilon-optimal
 * result reached in another order than the reference's); FGOICP_SCHEDULE_SERIAL keeps the reference's exact trajectory
 * (fgoicp.cpp:32-100: same pops, pushes, counters and result as the one-GPU SERIAL run, bit for bit, on every rank) and deals the
 * inner BnBs of every speculative evaluation over the ranks — one all-gather per evaluation, cooperative refinements. */
typedef struct fgoicp_multi fgoicp_multi;
enum { FGOICP_TRANSPORT_RCCL = 0, FGOICP_TRANSPORT_IN_PROCESS = 1 };
int fgoicp_multi_create(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, float lut_resolution, float mse_threshold,
                        const fgoicp_solver_opts* opts, const int* devices, int ndev, int transport, fgoicp_multi** out);
void fgoicp_multi_destroy(fgoicp_multi* m);
/* FastGoICP::run() on all ranks at once; R, t as fgoicp_solver_run (every rank ends with the same incumbent — checked).
 * If a rank fails, the exchange is aborted for all of them (nobody waits for the failed rank) and the call returns that rank's
 * status and message; with the RCCL transport the object cannot run again afterwards. */
int fgoicp_multi_run(fgoicp_multi* m, float* R_out9, float* t_out3);
/* One IterativeClosestPoint3D::run() (icp3d.cu:80-108) executed by ALL ranks together: each scans 1/world of the source for the two
 * exact nearest-neighbour passes of an iteration, the per-query results are all-gathered on device memory, sums and SVD are
 * replicated.  Same (sse, R, t, iterations) as fgoicp_icp on one GPU, bit for bit, on every rank (checked).  This is what every
 * refinement of a multi-rank run is (fgoicp_exchange.allgather_device). */
int fgoicp_multi_icp(fgoicp_multi* m, const float* R0, const float* t0, size_t max_iter, float convergence_threshold, float* sse_out, float* R_out9, float* t_out3,
                     int* iterations_out);
int fgoicp_multi_world(const fgoicp_multi* m);
fgoicp_solver* fgoicp_multi_solver(fgoicp_multi* m, int rank);     /* borrowed: stats, getters */
int fgoicp_multi_seconds(const fgoicp_multi* m, int rank, double* seconds);   /* wall-clock of that rank's last run() */
/* Scaling rehearsal on fewer GPUs than ranks: record what every exchange returned during a run, then run ONE rank alone
 * against the recording — the time that rank would need on a GPU of its own, without the collectives' latency. */
int fgoicp_multi_set_record(fgoicp_multi* m, int on);
/* What the last recorded run exchanged: the host-side collectives of `rank` and the device all-gathers of the cooperative ICP runs. */
int fgoicp_multi_recorded(const fgoicp_multi* m, int rank, uint64_t* host_exchanges, uint64_t* device_allgathers);
/* TEST HOOK, not part of the drop-in surface: the call-th exchange of `rank` in the next run fails, once. */
int fgoicp_multi_test_fault(fgoicp_multi* m, int rank, long call);
int fgoicp_multi_replay_rank(fgoicp_multi* m, int rank, double* seconds_out);

#ifdef __cplusplus
}
#endif
#endif /* FGOICP_AMD_H */
