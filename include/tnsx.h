/*
 * tnsx.h -- C ABI of the MI355X-native fixed-radius neighbour-search engine.
 *
 * This is the drop-in boundary for the `tns::TreeNSearch` hot path of
 * InteractiveComputerGraphics/TreeNSearch.  Every entry point replaces one member of the reference's
 * public class (reference paths relative to the upstream repository root):
 *
 *   TreeNSearch/source/TreeNSearch.h:28-427     class tns::TreeNSearch            (API surface)
 *   TreeNSearch/source/TreeNSearch.cpp:20-261   thin setters / getters
 *   TreeNSearch/source/TreeNSearch.cpp:138-149  run()  -> tnsx_run
 *   TreeNSearch/source/TreeNSearch.cpp:2571     prepare_zsort() -> tnsx_prepare_zsort
 *   TreeNSearch/source/NeighborList.h:8-39      handle layout [count, j0, j1, ...] -> tnsx_csr_view
 *
 * The header-only C++ shim include/tns/TreeNSearch.h re-creates the reference class verbatim on top of
 * this ABI; treensearch_amd/api.py is the ctypes mirror used by the tests.
 *
 * Conventions: plain pointers and sizes only; every function returns a tnsx_status (0 = ok) unless it
 * returns an id or a count (then negative = -status); tnsx_last_error() gives the message.  The engine
 * NEVER falls back to a CPU path: without a usable gfx950 device tnsx_create fails.
 */
#ifndef TNSX_H
#define TNSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNSX_VERSION 600

typedef struct tnsx_context tnsx_context;

typedef enum tnsx_status {
	TNSX_OK = 0,
	TNSX_ERR_INVALID = 1,      /* bad argument / API misuse (the reference prints + exit(-1), e.g. TreeNSearch.cpp:22-25) */
	TNSX_ERR_NO_DEVICE = 2,    /* no usable HIP device */
	TNSX_ERR_HIP = 3,          /* a HIP runtime call failed */
	TNSX_ERR_CONFIG = 4,       /* _check() failures, TreeNSearch.cpp:366-392 */
	TNSX_ERR_GRID_TOO_LARGE = 5, /* > 32768 cells per dimension, TreeNSearch.cpp:510-515 */
	TNSX_ERR_LIST_TOO_LONG = 6,
	TNSX_ERR_STATE = 7,        /* e.g. view requested before run(), pair inactive */
	TNSX_ERR_TIMEOUT = 8       /* slab layer watchdog: the stream did not drain in time (a mismatched or stuck exchange) */
} tnsx_status;

/* Distance arithmetic (both are bit-exact restatements of a reference build):
 *   STRICT      d2 = ((dx*dx + dy*dy) + dz*dz), every op rounded -- TreeNSearch.cpp:2478-2483 /
 *               BruteforceNSearch.cpp:88 as written (reference compiled with -ffp-contract=off)
 *   CONTRACTED  d2 = fma(dz,dz, fma(dx,dx, dy*dy)) -- what GCC emits for the same lines under the
 *               reference's own flags (-O3 -march=native, default -ffp-contract=fast) */
typedef enum tnsx_arith {
	TNSX_ARITH_STRICT = 0,
	TNSX_ARITH_CONTRACTED = 1
} tnsx_arith;

/* flags of tnsx_add_point_set / tnsx_resize_point_set */
#define TNSX_F32        0u   /* const float*  xyzxyz.. (+ const float* radii)   TreeNSearch.h:50,112 */
#define TNSX_F64        1u   /* const double* xyzxyz.. (+ const double* radii)  TreeNSearch.h:63,126 */
#define TNSX_HOST       0u   /* pointers are host memory; re-read (H2D) at every tnsx_run, TreeNSearch.h:375-378 */
#define TNSX_DEVICE     2u   /* pointers are device (HBM) memory; read in place at every tnsx_run */
#define TNSX_VARIABLE   4u   /* variable-radius set even though radii == NULL (legal only with n_points == 0;
                                the reference's tests hand null pointers for empty sets, tests.cpp:453) */

typedef struct tnsx_options {
	int device_id;            /* HIP device ordinal; -1 = current device */
	void* stream;             /* hipStream_t to launch on; NULL = the engine creates its own */
	int arith;                /* tnsx_arith; default STRICT */
	int mirror_to_host;       /* 1: tnsx_run also mirrors every active pair's lists into pinned host memory
	                             (what get_neighborlist needs for CPU consumers); 0: lists stay in HBM */
	int collect_stage_times;  /* 1: record hipEvents around every stage (tnsx_get_stats) */
	int exact_layout;         /* 0 (default): the lists of a pair are built in ONE pass into a record pool sized from the
	                             previous run (per-wave slabs from a device cursor; records exact and contiguous, order of
	                             records in memory unspecified, pool has unused gaps; the first run of a pair adds a dry,
	                             count-only pass).  1: always count -> scan -> fill, records laid out in spatially sorted
	                             point order without gaps (deterministic, ~2x more query work) */
	uint64_t max_dense_cells; /* upper bound of the dense cell table (8 bytes per cell and point set), at most 2^30.  0 = default:
	                             max(2^22, 64 x the number of points) cells -- a sparse scene (one stray particle far away) gets coarser
	                             cells, which is exact but slower for the affected cells, instead of a table of gigabytes */
	int temporal_reuse;       /* 1 (default): a run lays the previous run's search grid over the points without computing their bounds
	                             first, and point sets whose input did not change keep their sorted arrays and cell table; both are
	                             verified on the device during the run and the run is repeated when an assumption was wrong (the
	                             reference's own reuse, TreeNSearch.cpp:474-482 and :77-79).  0: bounds and full build every run.
	                             Guarantee: grid reuse is verified exactly (every point is tested against the box).  "Input did not
	                             change" is decided by pointer + size + a 64-bit order-sensitive checksum of the set's raw bits that
	                             the device recomputes every run: a changed set whose checksum collides (probability ~2^-64 per run
	                             for unrelated inputs) would keep stale lists.  Callers that need a deterministic guarantee set 0 */
	int sorted_lists;         /* 1: every neighbour list is put into ascending index order after the query (one wave per record, bitonic
	                             network), as the reference's lists are by construction (TreeNSearch.cpp:2474-2500) -- sums over
	                             neighbours are then evaluated in the same order as on the CPU.  0 (default): order unspecified */
	int n_devices;            /* > 1: multi-device mode for HOST-resident inputs (the C++ drop-in): every run is cut into that many slabs
	                             along x (balanced cuts from an x histogram, one-halo ghosts added to each slab's upload), one engine per
	                             device, lists gathered into one pinned host buffer -- N PCIe links instead of one for the copy that
	                             bounds drop-in mode.  Device pointers, device views and the tnsx_halo_pack family are not available on
	                             such a context.  0 / 1: single device (device_id) */
	int device_ids[8];        /* HIP device ordinals of the n_devices engines (an ordinal may repeat: several engines on one GPU) */
	int query_blocks_per_cu;  /* tuning: workgroups per CU of the general query kernel (1..16); 0 = default (7) */
	int fast_blocks_per_cu;   /* tuning: workgroups per CU of the fast pool kernels (1..16); 0 = default (8).  Both are fixed at tnsx_create:
	                             nothing in the launch path reads the environment */
	int bucket_build_min_points; /* sets with at least this many points are built with the two-pass bucket build where
	                             their key allows it; 0 = default (65536), < 0: never (always the stable LSD passes + k_cell_table) */
	int sparse_grid;          /* 0 = default: a grid too large for a dense cell table (a cloud that is sparse everywhere: a sheet, a filament) keeps cells of one
	                             search radius as a SPARSE grid -- occupied-cell lists + block index, up to 2^32 cells -- instead of coarser cells; < 0: never
	                             (coarsen the cells until a dense table fits, the behaviour of rounds 1-3) */
	int query_formulation;    /* 0 = default: the cell kernels (candidates in the lanes).  1 = experiment of round 3, measured
	                             SLOWER (profiles/r3_group_formulation.txt) and NOT part of libtnsx.so (tnsx_query_formulation_available answers 0; its
	                             source is tools/ubench/tnsx_query_group.hip, tools/build_group_variant.sh builds a variant of the library that carries
	                             it): a fixed-radius search of a set in itself first runs the group formulation --
	                             16 query points of a cell per batch in the lanes, the tests as 16x16x4 fp32 MFMAs with an exact re-test
	                             inside the rounding band -- and the cell kernels take the cells it passes on; a pair that passes on more
	                             than a quarter of its cells goes back to the cell kernels alone.  Results are identical either way */
} tnsx_options;

/* Neighbour lists of one active (set_i -> set_j) pair.  Record layout == the reference's chunk storage
 * (vectors_internals.h:152-174, TreeNSearch.cpp:2494-2500): records[offsets[p]] = n, followed by the n
 * set-local indices into set_j, so `tns::NeighborList(records + offsets[p])` works unchanged. */
typedef struct tnsx_csr_view {
	int n_points;                    /* points in set_i */
	uint64_t n_records;              /* ints of `records` in use: total neighbours + n_points (+ unused gaps in pool mode) */
	uint64_t n_neighbors;            /* total neighbour indices */
	const uint64_t* offsets_device;  /* [n_points] by ORIGINAL point index, HBM */
	const int* records_device;       /* [n_records], HBM */
	const uint64_t* offsets_host;    /* pinned host mirror, NULL unless mirrored.  Round 5: the mirror is a GAP-FREE copy in point order -- n_neighbors + n_points ints,
	                                    records_host[offsets_host[p]] = count of point p followed by its indices, the record of point p + 1 right behind it -- so these
	                                    offsets are NOT the device offsets (the device records keep the pool's layout, holes included) .  A multi-device context
	                                    (tnsx_options.n_devices > 1) keeps one region of records per slab in its host view: there n_records ints are valid and
	                                    the offsets address them; read every list through its offset, never by walking the records front to back */
	const int* records_host;
} tnsx_csr_view;

typedef struct tnsx_stats {
	/* last tnsx_run */
	int n_sets;
	uint64_t n_points;            /* all sets */
	uint64_t n_queries;           /* Q: sum over active pairs of n_i */
	uint64_t n_neighbors;         /* E: total emitted indices */
	uint64_t n_occupied_cells;    /* C: summed over sets */
	uint64_t n_grid_cells;        /* cells of the search grid */
	int grid_dims[3];
	float grid_cell_size;
	float grid_origin[3];         /* the grid covers the tight bounds of the points widened by two search radii (never beyond the world box) */
	int key_bits, radix_passes;
	/* algorithmic HBM bytes of the last run (SURVEY.md section 8d formula, evaluated with measured Q,E,C) */
	uint64_t bytes_build, bytes_query;
	/* stage times in ms of the last attempt (0 unless collect_stage_times): H2D of host inputs; bounds kernels + their host round trip
	 * (0 when the grid was reused); clearing the previous run's cell-table entries; cell sort; cell table; count / scan passes
	 * (exact_layout only); the query pass(es) that write the lists; pinned host mirror */
	float ms_total, ms_upload, ms_bounds, ms_table_clear, ms_sort, ms_cells, ms_count, ms_scan, ms_fill, ms_mirror;
	float ms_sort_lists;          /* sorted_lists: the pass that orders every record */
	int n_pool_pairs;             /* pairs built in single-pass pool mode in the last run */
	int pool_retries;             /* pool passes repeated because the pool was too small */
	int cold_passes;              /* dry (count-only) passes of pairs that ran for the first time */
	int speculated;               /* 1: the last run reused the previous run's grid (no bounds pass, no host round trip before the build) */
	int speculation_redos;        /* attempts of the last run that were thrown away because an assumption was wrong (0 or 1) */
	int n_cached_sets;            /* point sets whose build was skipped in the last run (input unchanged) */
	uint32_t n_filtered_cells;    /* pairs of two different sets: query cells that have any candidate (the others share ONE empty record and
	                                 are never visited), summed over those pairs */
	int n_devices_used;           /* multi-device mode: slabs the last run was cut into (0 on a single-device context) */
	/* world box of the reference semantics (TreeNSearch.cpp:415-522) */
	float world_bottom[3], world_top[3];
	int world_cells_pow2;
	float zsort_cell_size_inv;    /* 1 / quantisation step of the last tnsx_prepare_zsort: 1 / cell size after a run() (cell-level order, the
	                                 reference's tree path), the cell size halved down to < 2^21 steps per axis otherwise (its no-tree path) */
	int grid_trimmed;             /* 1: cells of one search radius over the bounding box of all points would not fit (far outliers): the grid of the last
	                                 run covers the bulk of the points, the rest sits in its border cells (exact all the same) */
	uint32_t n_group_pairs;       /* pairs of the last run that ran the group formulation */
	uint32_t n_group_passed_cells; /* occupied cells it passed on to the cell kernels (too many candidates or query points, a list overflow, a point
	                                 far outside its cell), summed over those pairs */
	int grid_sparse;              /* 1: the grid of the last run had more cells than a dense table may have (tnsx_options.max_dense_cells); the cells kept their
	                                 edge of one search radius and were held as key-ordered lists of occupied cells with a block index instead (slower look-ups) */
	int one_read_builds;          /* point sets whose bucket build read the input once in the last run (windows from the previous run) */
	int heavy_catchups;           /* pool passes whose heavy tiers (cells with > 512 candidates or > 64 query points) were not launched with the first
	                                 tier -- the previous run of the pair had no such cell -- and had to run after the run's synchronisation */
	int sampled_passes;           /* count-only passes of the last run that looked at every 32nd occupied cell only (first run of a pair of a set of >= 2^20 points:
	                                 the pool is sized from the scaled counts; cold_passes counts the full ones) */
	int nan_fixups;               /* pool passes of the last run after which query points that entered no cell (NaN x: "no point") had their offsets pointed
	                                 at the pool's empty record (a pass whose records do not add up to neighbours + queries; never in a run without such points) */
} tnsx_stats;

/* ---- lifetime ---------------------------------------------------------------------------------- */
tnsx_status tnsx_default_options(tnsx_options* opt);
tnsx_status tnsx_create(const tnsx_options* opt /* may be NULL */, tnsx_context** out);   /* TreeNSearch.h:36 */
void        tnsx_destroy(tnsx_context* ctx);                                               /* TreeNSearch.h:37 */
const char* tnsx_last_error(const tnsx_context* ctx /* NULL: creation errors */);
int         tnsx_version(void);
int         tnsx_get_device(const tnsx_context* ctx);   /* HIP device ordinal the context's memory and stream live on (multi-device: the first slab's) */
/* 1 when this build of the library carries the query formulation `f` of tnsx_options.query_formulation (0: always; 1: only when the library was
 * built with -DTNSX_WITH_GROUP_FORMULATION by tools/build_group_variant.sh -- never the product library), else 0.  Asking a context for a formulation its library does not carry is
 * not an error: the run uses formulation 0 (the results are identical by contract). */
int         tnsx_query_formulation_available(int f);

/* ---- point sets (TreeNSearch.cpp:35-133, 346-365) ---------------------------------------------- */
/* returns the set id (>= 0) or -status.  radii == NULL => fixed-radius mode set. */
int         tnsx_add_point_set(tnsx_context* ctx, const void* xyz, const void* radii, int n_points, unsigned flags);
tnsx_status tnsx_resize_point_set(tnsx_context* ctx, int set_id, const void* xyz, const void* radii, int n_points,
                                  unsigned flags);

/* ---- configuration ------------------------------------------------------------------------------ */
tnsx_status tnsx_set_search_radius(tnsx_context* ctx, float r);          /* TreeNSearch.cpp:20-34 */
tnsx_status tnsx_set_cell_size(tnsx_context* ctx, float cell_size);      /* TreeNSearch.cpp:173-182 (write-once) */
tnsx_status tnsx_set_symmetric_search(tnsx_context* ctx, int active);    /* TreeNSearch.cpp:169 */
tnsx_status tnsx_set_active_search(tnsx_context* ctx, int set_i, int set_j, int active);            /* :219-222 */
tnsx_status tnsx_set_active_search_all(tnsx_context* ctx, int set_i, int search_in_all, int be_found_by_all); /* :223-232 */
tnsx_status tnsx_set_all_searches(tnsx_context* ctx, int active);        /* TreeNSearch.cpp:233-240 */
tnsx_status tnsx_set_arithmetic(tnsx_context* ctx, int arith);           /* tnsx_arith */
tnsx_status tnsx_set_collect_stage_times(tnsx_context* ctx, int on);     /* tnsx_options.collect_stage_times from the next run on (every event record
                                                                            between two kernels costs a bubble of 6-9 us: off in timed loops) */

/* ---- getters (TreeNSearch.cpp:191-218) ---------------------------------------------------------- */
int         tnsx_get_n_sets(const tnsx_context* ctx);
int         tnsx_get_n_points_in_set(const tnsx_context* ctx, int set_i);
int64_t     tnsx_get_total_n_points(const tnsx_context* ctx);
int         tnsx_is_search_active(const tnsx_context* ctx, int set_i, int set_j);
int         tnsx_does_set_exist(const tnsx_context* ctx, int set_i);
uint64_t    tnsx_get_neighborlist_n_bytes(const tnsx_context* ctx);      /* TreeNSearch.cpp:254-261 */

/* ---- the hot path ------------------------------------------------------------------------------- */
/* run(): _set_up, _check, world box, build + query of every active pair (TreeNSearch.cpp:138-149).
 * Synchronous like the reference: when it returns the lists are complete (in HBM, and in pinned host
 * memory when mirror_to_host). */
tnsx_status tnsx_run(tnsx_context* ctx);
/* run_scalar() (TreeNSearch.cpp:150-160).  Same device path and the same neighbour sets as tnsx_run -- the reference's scalar
 * kernel accumulates the distance in double (TreeNSearch.cpp:2080-2087) and is not a parity target, see DESIGN.md section 2 --
 * but the WORLD BOX follows _update_world_AABB (:415-522: the tight box), where tnsx_run follows _update_world_AABB_simd
 * (:523-645: the tight box united with the origin, an artefact of its zero-padded remainder loop :564-569). */
tnsx_status tnsx_run_scalar(tnsx_context* ctx);
/* get_neighborlist() backing store (TreeNSearch.cpp:241-249).  Views stay valid until the next tnsx_run. */
tnsx_status tnsx_get_pair_view(tnsx_context* ctx, int set_i, int set_j, tnsx_csr_view* out);
/* mirrors one pair to pinned host memory on demand (no-op when already mirrored) */
tnsx_status tnsx_mirror_pair_to_host(tnsx_context* ctx, int set_i, int set_j);
/* copies one pair's offsets / records into caller memory (host or device pointers, either may be NULL) */
tnsx_status tnsx_copy_pair(tnsx_context* ctx, int set_i, int set_j, uint64_t* offsets_dst, int* records_dst,
                           int dst_on_device);
/* The pair as a standard, gap-free CSR in point order, built on the device into caller-provided DEVICE memory ("CSR tensors
 * out" -- what get_neighborlist, TreeNSearch.cpp:241-249, hands out one point at a time, for consumers that live on the GPU): offsets_out[p] ..
 * offsets_out[p + 1] delimit the neighbours of point p in indices_out; n_points + 1 offsets, n_neighbors indices (tnsx_get_pair_view gives both
 * numbers), no count words; the order inside a list is the records' order.  indices_out may be NULL (offsets = running neighbour counts only).
 * Returns when the arrays are complete.  Single-device contexts only. */
tnsx_status tnsx_pair_csr_device(tnsx_context* ctx, int set_i, int set_j, int64_t* offsets_out, int* indices_out);

/* ---- z-sort (TreeNSearch.cpp:2571-2716, TreeNSearch.h:443-481) ---------------------------------- */
tnsx_status tnsx_prepare_zsort(tnsx_context* ctx);
/* new -> old map of one set (TreeNSearch.cpp:250-253); host copy and device copy, valid until the next prepare */
tnsx_status tnsx_get_zsort_order(tnsx_context* ctx, int set_i, const int** new_to_old_host,
                                 const int** new_to_old_device, int* n);
/* data[new*stride + s] = old_data[old*stride + s] for elements of elem_bytes bytes, in place
 * (apply_zsort<T>, TreeNSearch.h:443-481); data may be host or device memory */
tnsx_status tnsx_apply_zsort(tnsx_context* ctx, int set_i, void* data, size_t elem_bytes, int stride, int on_device);

/* ---- introspection ------------------------------------------------------------------------------- */
tnsx_status tnsx_get_stats(const tnsx_context* ctx, tnsx_stats* out);

/* ---- multi-GPU support (no counterpart in the single-process reference) ----------------------
 * The slab layer (treensearch_amd/multi.py: one process per GPU, slabs along x, one halo exchange per step over RCCL) is built
 * from these four device-side pieces; all pointers are device memory and everything runs on the context's stream. */

/* Ghost-halo selection: packs every point with x < left_cut into out_left and every point with x >= right_cut into out_right
 * (either may be NULL = side not wanted), as rows of `5 + (radii != NULL)` floats: x, y, z, [r,] and the point's 64-bit global
 * id bit-cast into the last two floats.  Rows are appended in no particular order.  counts_dev[0..1] (device scratch) and
 * counts_host[0..1] receive the number of rows the selection HAS per side (they may exceed that side's capacity: then only
 * capacity rows were written and the caller repeats with a larger buffer).  With counts_host != NULL the call waits for the
 * stream, with NULL it only enqueues.  Like tnsx_run, it does not wait for work other streams still have in flight on its inputs. */
tnsx_status tnsx_halo_pack(tnsx_context* ctx, const float* xyz, const float* radii, const long long* global_ids, int n_points,
                           float left_cut, float right_cut, float* out_left, float* out_right, unsigned long long capacity_left,
                           unsigned long long capacity_right, unsigned int* counts_dev, unsigned int* counts_host);
/* Balanced slab cuts: hist_dev[clamp(trunc((x - x0) * inv_dx), 0, n_bins - 1)] += 1 for every point (the caller zeroes hist_dev,
 * all-reduces it over the ranks and cuts at the quantiles).  Enqueues only. */
tnsx_status tnsx_x_histogram(tnsx_context* ctx, const float* xyz, int n_points, float x0, float inv_dx, int n_bins,
                             unsigned int* hist_dev);
/* Candidates-only tail: from the next tnsx_run on, only the first n_query points of set_i get neighbour lists; the others are
 * still found as neighbours (the ghost points a slab appends to its owned points).  n_query < 0: all points (default).  The views
 * of pairs (set_i -> *) then hold n_points = min(n, n_query) records; offsets of the remaining points are unspecified. */
tnsx_status tnsx_set_query_count(tnsx_context* ctx, int set_i, int n_query);
/* User ids: from the next tnsx_run on, the lists of every pair (* -> set_i) hold ids_dev[j] instead of the index j (the global ids
 * of [owned | ghosts] of a slab, so that nothing is left to translate).  ids_dev: device array with one int per point of the set,
 * re-read at every run like the coordinates; NULL switches back to indices.  Ids of one set must be distinct when the set is
 * searched in itself.  Costs one 4-byte gather and one 4-byte store per point in the last pass of the cell sort. */
tnsx_status tnsx_set_point_ids(tnsx_context* ctx, int set_i, const int* ids_dev);
/* waits for everything the context has enqueued on its stream (tnsx_halo_pack / tnsx_x_histogram with no host result) */
tnsx_status tnsx_synchronize(tnsx_context* ctx);
/* Rewrites every neighbour index j of pair (set_i -> set_j) as id_map_dev[j], in place in HBM (local -> global ids of
 * [owned | ghosts]); id_map_dev must have one entry per point of set_j.  Waits for completion. */
tnsx_status tnsx_translate_neighbors(tnsx_context* ctx, int set_i, int set_j, const int* id_map_dev);

/* ---- slab layer: the multi-GPU path behind the C ABI (tnsx_slab.cpp) ---------------------------------------
 * One tnsx_slab per GPU (one process per GPU, or one thread per GPU) on top of an ordinary single-device context.  Rank k owns the
 * points with slab_lo <= x < slab_hi; tnsx_slab_step() sends the points within one halo width of a slab face to that neighbour
 * (tnsx_halo_pack + ONE grouped ncclSend / ncclRecv per step over RCCL), appends the received ghosts to the owned points of their set
 * as candidates-only points with their global ids, and runs the engine: the lists of the owned points then hold GLOBAL ids.  The
 * lists are read through the engine as usual (tnsx_get_pair_view(engine, tnsx_slab_engine_set(slab, i), tnsx_slab_engine_set(slab, j))).
 * Steady state is speculative (fixed-capacity messages, nothing read on the host before the search, capacities checked afterwards);
 * a link whose capacity was exceeded is repaired by its two ends alone and only they search again. */
typedef struct tnsx_slab tnsx_slab;

/* one message pair with one neighbour: send_bytes from `send` to rank `peer`, recv_bytes from `peer` into `recv` (device memory; 0 bytes: no message) */
typedef struct tnsx_slab_op { int peer; const void* send; size_t send_bytes; void* recv; size_t recv_bytes; } tnsx_slab_op;
enum { TNSX_SLAB_SUM_U32 = 0, TNSX_SLAB_MIN_F32 = 1, TNSX_SLAB_MAX_F32 = 2 };
/* How the messages move.  tnsx_slab_transport_rccl fills it for RCCL; tnsx_slab_transport_local for slabs that live in one process
 * (tests on a single GPU); an application with its own communication layer (MPI, ...) fills it itself.  Both calls order their work
 * after what `stream` (a hipStream_t) already holds and must be complete, or ordered on `stream`, when they return. */
typedef struct tnsx_slab_transport {
	void* user;
	int (*exchange)(void* user, int rank, int world, const tnsx_slab_op* ops, int n_ops, void* stream);   /* all ops of one round, 0 = ok, 3 = timed out
	                                                                                                        (reported as TNSX_ERR_TIMEOUT), else failed */
	int (*allreduce)(void* user, int rank, int world, void* dev_buf, int count, int op, void* stream);    /* in place, 32-bit elements */
	void (*release)(void* user);
	void (*abort)(void* user);   /* may be NULL.  Called by the watchdog when an exchange did not complete in time: must make the pending operations of
	                                this rank return (RCCL: ncclCommAbort), so that the process can report the error and leave instead of hanging */
} tnsx_slab_transport;
typedef struct tnsx_slab_info {
	int n_owned, n_ghost;        /* set 0 of the last step */
	int speculative_last;        /* 1: the last step was one round of fixed-capacity messages */
	int redone_last;             /* 1: a capacity was exceeded, the overflowed link(s) were repaired and the search repeated */
	int rounds_last;             /* exchange rounds of the last step (exact step: 2, speculative: 1, + 1 repair round) */
	unsigned long long bytes_sent;   /* so far */
	int transport_kind;          /* 0: none (one slab), 1: RCCL (tnsx_slab_transport_rccl), 2: in-process (tnsx_slab_transport_local), 3: the application's own */
	int transport_ranks;         /* ranks the transport spans as far as the library can tell: ncclCommCount of the RCCL communicator, the size of a local
	                                group, 1 without a transport, -1 for an application's transport (tnsx_slab_transport_check counts those) */
	float exchange_ms_last;      /* tnsx_slab_set_collect_times(1): time the exchange rounds of the last step took on the stream (hipEvent pairs around the
	                                transport's exchange calls; a speculative step has one round), else 0 */
} tnsx_slab_info;

/* RCCL: rank 0 makes the 128-byte id, the application hands it to every rank (MPI_Bcast, torch.distributed.broadcast, a file, ...) */
tnsx_status tnsx_slab_rccl_unique_id(void* out128);
tnsx_status tnsx_slab_transport_rccl(const void* unique_id128, int rank, int world, int device /* -1: current */, tnsx_slab_transport* out);
const char* tnsx_slab_rccl_error(void);
/* all slabs inside one process */
tnsx_status tnsx_slab_local_group_create(int world, void** group_out);
void        tnsx_slab_local_group_release(void* group);
tnsx_status tnsx_slab_transport_local(void* group, int rank, tnsx_slab_transport* out);
void        tnsx_slab_transport_release(tnsx_slab_transport* t);

/* Balanced cuts: global x range (all-reduce min / max), one histogram of the x planes of width plane_width (>= the halo, so that only
 * adjacent slabs ever exchange ghosts) per rank, all-reduce(sum), cuts at the plane boundaries closest to the k / n_slabs quantiles.
 * cuts_out[0 .. n_slabs]: cuts_out[0] = -inf, cuts_out[n_slabs] = +inf; slab k owns cuts[k] <= x < cuts[k + 1].  Collective: every
 * rank calls it with its own points (device memory).  n_slabs <= 0: one slab per rank. */
tnsx_status tnsx_slab_balanced_cuts(tnsx_context* engine, const tnsx_slab_transport* transport, int rank, int world, int n_sets,
                                    const float* const* xyz, const int* n_points, float plane_width, int n_slabs, float* cuts_out);

/* radius > 0: fixed search radius of all sets (set on the engine here); radius <= 0: per-point radii, max_radius must bound every
 * radius of every rank (it sizes the halo; checked on the owned radii at every step).  halo_margin: the halo is
 * max_radius * (1 + halo_margin) wide (<= 0: 1e-3).  transport may be NULL when world == 1. */
/* The one all-to-all of a decomposition (SURVEY.md section 8e "Exchange"): every point moves to the slab that owns its x, slab k owning
 * cuts[k] <= x < cuts[k + 1] (cuts[0 .. world] as tnsx_slab_balanced_cuts writes them; world <= 64).  Collective, two rounds over the transport's
 * exchange (the counts, then the rows [x, y, z, (r,) gid] between every pair of ranks; a rank's own share is a device copy).
 * _begin: xyz / gids / radii (radii may be NULL) are this rank's n_points points in device memory; *n_owned = the points this rank owns afterwards.
 * _finish: writes them to caller-allocated device arrays of n_owned points (order unspecified) and frees the handle; NULL outputs: just free. */
typedef struct tnsx_slab_redist tnsx_slab_redist;
tnsx_status tnsx_slab_redistribute_begin(tnsx_context* engine, const tnsx_slab_transport* transport, int rank, int world, const float* cuts,
                                         const float* xyz, const long long* gids, const float* radii, int n_points, tnsx_slab_redist** out, int* n_owned);
tnsx_status tnsx_slab_redistribute_finish(tnsx_slab_redist* r, float* xyz_out, long long* gids_out, float* radii_out);
/* bound of the two waits of tnsx_slab_redistribute_begin (default 120 s; <= 0: for ever; process-wide).  On expiry the transport's abort is called when it has one;
 * when it has none the buffers the stream may still touch are leaked rather than freed under it. */
tnsx_status tnsx_slab_set_redistribute_watchdog(double seconds);

/* A slab with two neighbours must be at least one halo wide (ghosts come from the adjacent slabs only): TNSX_ERR_INVALID otherwise.
 * Creation errors are described by tnsx_slab_last_error(NULL).  The engine must outlive the slab: tnsx_slab_destroy turns the engine's sets that
 * point into the slab's buffers into empty sets before it frees them. */
tnsx_status tnsx_slab_create(tnsx_context* engine, const tnsx_slab_transport* transport, int rank, int world, float slab_lo, float slab_hi,
                             float radius, float max_radius, float halo_margin, int speculative, tnsx_slab** out);
void        tnsx_slab_destroy(tnsx_slab* slab);
const char* tnsx_slab_last_error(const tnsx_slab* slab /* NULL: the last creation / decomposition error of this thread */);
/* Watchdog: every wait of tnsx_slab_step on the stream (the exchange, the search behind it) is bounded by `seconds` (default 120; <= 0: wait for
 * ever).  When it expires the step names the link(s) it was waiting on in tnsx_slab_last_error, calls the transport's abort and returns
 * TNSX_ERR_TIMEOUT.  What is bounded: the waits on the STREAM, and the host-side waits of the in-process transport (tnsx_slab_transport_local); an application's
 * own transport bounds its own host-side waits (return 3 from exchange) -- a mismatched exchange on 8 GPUs fails with a message instead of hanging the job. */
tnsx_status tnsx_slab_set_watchdog(tnsx_slab* slab, double seconds);
/* searches between the slab's sets (indices as in tnsx_slab_step); default: set 0 in itself */
tnsx_status tnsx_slab_set_active_search(tnsx_slab* slab, int set_i, int set_j, int active);
/* One step: exchange + search.  Per set k: xyz[k] (n_points[k] x 3 floats), gids[k] (global ids, must fit 31 bits: they become the
 * int indices of the lists), radii[k] (per-point radii mode only) -- device memory, the OWNED points of this rank. */
tnsx_status tnsx_slab_step(tnsx_slab* slab, int n_sets, const float* const* xyz, const long long* const* gids, const float* const* radii,
                           const int* n_points);
int         tnsx_slab_engine_set(const tnsx_slab* slab, int set_index);   /* the engine's id of [owned | ghosts] of that set, -1 before its first step */
tnsx_status tnsx_slab_get_info(const tnsx_slab* slab, tnsx_slab_info* out);
/* 1: from the next step on every exchange round is bracketed by a pair of hipEvents on the stream (tnsx_slab_info.exchange_ms_last).  Off by default: an
 * event record between two launches costs a bubble of several microseconds -- benchmarks switch it on for extra steps behind their timed loop. */
tnsx_status tnsx_slab_set_collect_times(tnsx_slab* slab, int on);
/* Collective self check of a transport: all-reduces the number 1 over its ranks on the engine's stream and returns the sum -- `world` when every rank of
 * the job really is on the other end (a benchmark line can then say that N ranks took part without anybody having to trust the launcher). */
tnsx_status tnsx_slab_transport_check(tnsx_context* engine, const tnsx_slab_transport* transport, int rank, int world, int* ranks_seen);
tnsx_status tnsx_slab_debug_set_capacity(tnsx_slab* slab, int side, unsigned rows);   /* tests: shrink the agreed capacity of one link (0 = left) */

#ifdef __cplusplus
}
#endif
#endif /* TNSX_H */
