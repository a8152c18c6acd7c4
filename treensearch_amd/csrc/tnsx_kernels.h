// Internal launcher interface between the host engine (tnsx_engine.cpp) and the gfx950 kernels
// (tnsx_kernels.hip).  Not part of the public ABI (that is include/tnsx.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace tnsx {

// Search grid: cells of edge h >= r_max (with an fp-rounding safety margin), row-major keys with x fastest:
// key = (iz*ny + iy)*nx + ix, so the three x-neighbours of a row are contiguous in sorted order.
struct GridParams {
	float ox, oy, oz;   // origin = tight minimum of all points
	float inv_h;        // 1/h (fp32, rounded)
	int nx, ny, nz;
};

// ---- elementwise ------------------------------------------------------------------------------
void launch_f64_to_f32(const double* in, float* out, size_t n, hipStream_t s);

// ---- bounds: {min xyz, max xyz, min r, max r} ---------------------------------------------------
int  bounds_num_blocks(int n);
void launch_bounds_partial(const float* xyz, const float* radii, int n, float* partials /*[nb*8]*/, hipStream_t s);
void launch_bounds_final(const float* partials, int n_partials, float* out8, hipStream_t s);

// ---- exclusive scans ----------------------------------------------------------------------------
size_t scan_temp_bytes(size_t n);
void exclusive_scan_u32_to_u64(const uint32_t* in, uint64_t* out, size_t n, void* temp, hipStream_t s);   // out[n+1], out[n]=total

// ---- build of the search structure of one point set (tnsx_build.hip) ---------------------------------------
// Cell sort: LSD radix sort on the cell key that moves the point itself, (x, y, z, bits(original index)) [+ r*r]; keys are
// recomputed from the positions, digits have up to CS_MAX_BITS bits.  xyzi / r2 ping-pong between [0] and [1];
// launch_cell_sort returns the index that holds the sorted points.
static constexpr int CS_MAX_BITS = 11;
struct CellSortPlan { int passes; int bits[8]; };
CellSortPlan cell_sort_plan(int key_bits);
struct CellSortBuffers { float4* xyzi[2]; float* r2[2]; };
size_t cell_sort_temp_bytes(int n);
// What a run of the engine speculates on, validated by the first pass of the sort (see tnsx_build.hip): flag (or nullptr) is
// raised when a point lies outside [lo, hi] or a radius exceeds r_max; checksum (or nullptr, must be zeroed) receives the
// order-sensitive 64-bit checksum of the set's points (+ radii), as CHK_SLOTS partial sums (see above).
// the checksum of a set is the sum of CHK_SLOTS partial sums that live CHK_STRIDE 64-bit words (one 128-byte line) apart
static constexpr int CHK_SLOTS = 64, CHK_STRIDE = 16;
struct BuildGuard {
	float lo[3] = { 0.f, 0.f, 0.f }, hi[3] = { 0.f, 0.f, 0.f };
	float r_max = 3.402823466e+38f;
	uint32_t* flag = nullptr;
	unsigned long long* checksum = nullptr;
	// a grid that covers the bulk of the points only (trim_box): lo / hi above are the WORLD box, and the points outside the grid's
	// own box [soft_lo, soft_hi] are counted -- they are binned into its border cells, which is exact but slow if they become many
	float soft_lo[3] = { 0.f, 0.f, 0.f }, soft_hi[3] = { 0.f, 0.f, 0.f };
	unsigned long long* outside = nullptr;
};
// ids != nullptr (tnsx_set_point_ids): the sorted points carry ids[original index] instead of the original index (that is what
// the query emits), and orig_sorted[sorted position] receives the original index.
// sort + cell table + occupied-cell list in one call: the two-pass bucket build where the key allows it (<= 24 bits, tnsx_build.hip),
// else launch_cell_sort + launch_cell_table.  query_limit: points with original index >= query_limit are candidates only and must
// come behind the others inside a cell; stable_order: keep the input order inside a cell (exact layout).  temp: cell_build_temp_bytes(n).
size_t cell_build_temp_bytes(int n);
// The one-read form of the bucket build's first pass (round 4, tnsx_build.hip k_bucket_scatter): every bucket has a window {first slot, capacity} of the
// intermediate array xyzi[1] -- written by the previous build of the same set on the same grid (win, n_buckets entries; every build writes the next
// run's) -- and a cursor (cursors[b * BUCKET_CURSOR_STRIDE], zero at the start of the run: launch_run_begin).  use = false: the histogram pass.
// xyzi[1] must hold bucket_window_slots(n, n_buckets) points whenever win is given.
static constexpr int BUCKET_CURSOR_STRIDE = 32;   // 128 bytes: returning atomics on one cache line serialise
struct BucketWindows { bool use = false; uint2* win = nullptr; uint32_t* cursors = nullptr; };
// Sparse grid (round 4): no dense table; occ becomes the key-ordered list of occupied cells with a sentinel behind it, blk its block index
// (n_blocks + 1 entries, one per 2^shift keys).  blk == nullptr: the dense table.
struct SparseCells { uint32_t* blk = nullptr; int shift = 0; uint32_t n_blocks = 0; };
bool cell_build_uses_buckets(int n, int key_bits, bool stable_order, int bucket_min_points, int* n_buckets);
size_t bucket_window_slots(int n, int n_buckets);
int launch_cell_build(const float* xyz, const float* radii, int n, GridParams g, int key_bits, const CellSortBuffers& b, void* temp, const int* ids,
                      uint32_t* orig_sorted, const BuildGuard& gd, uint32_t query_limit, bool stable_order, int bucket_min_points, uint2* table, uint2* occ,
                      uint32_t* n_occ, int* passes_out, const BucketWindows& bw, const SparseCells& sp, hipStream_t s);
int launch_cell_sort(const float* xyz, const float* radii, int n, GridParams g, int key_bits, const CellSortBuffers& b, void* temp, const int* ids,
                     uint32_t* orig_sorted, const BuildGuard& gd, hipStream_t s);
// the same checksum on its own (sets whose build is skipped)
void launch_set_checksum(const float* xyz, const float* radii, int n, unsigned long long* out, hipStream_t s);
// The same sort on the Morton code of the point's cell on the REFERENCE grid (TreeNSearch.cpp:713-715 quantisation, libmorton bit
// order; prepare_zsort): g.ox/oy/oz = world bottom, g.inv_h = 1 / cell size, g.nx = cells per axis (a power of two), 3 * log2(nx)
// key bits.  order_out[p] = original index of the p-th point in z-order.
int launch_morton_sort(const float* xyz, int n, GridParams g, int key_bits, const CellSortBuffers& b, void* temp, int* order_out, hipStream_t s);
size_t zsort_temp_bytes(int n, int key_bits);   // temp of launch_morton_sort: max(cell_sort_temp_bytes(n), this)
// Cell table: table[key] = (first sorted position, one past last); occ = {first sorted position, key} of every occupied cell
// (order of blocks of 4096 points is arbitrary), *n_occ = their number (must be zeroed before)
void launch_cell_table(const float4* xyzi_sorted, int n, GridParams g, uint2* table, uint2* occ, uint32_t* n_occ, hipStream_t s);
// zeroes the table entries of the first n_occ cells of an occupied-cell list (instead of a memset of the whole table)
void launch_table_clear(const uint2* occ, uint32_t n_occ, uint2* table, hipStream_t s);

// ---- the query ----------------------------------------------------------------------------------
struct QueryArgs {
	// query set i
	const uint2* occ_i; const uint32_t* n_occ_i; const uint2* table_i;
	const float4* xyzi_i; const float* r2_i;
	const uint32_t* orig_i;   // original index by sorted position of set i, or nullptr: it is the w component of xyzi_i (no user ids)
	// candidate set j
	const uint2* table_j; const float4* xyzi_j; const float* r2_j;
	float r2_fixed;
	uint32_t shared_empty;  // pool pass: offs_by_orig was pre-set to 0 and records[0] == 0 is THE empty record (cells without candidates write nothing)
	uint32_t n_points_i;    // points of set i (length of xyzi_i; group formulation)
	uint32_t query_limit;   // only query points with original index < query_limit get lists (the rest of set i are candidates only)
	GridParams g;
	// count pass: counts[p] = n_neighbours + 1 (record length), by sorted position of set i
	uint32_t* counts;
	// fill pass
	const uint64_t* offs_sorted;   // exclusive scan of counts
	int* records;                  // [count, j...] records
	uint64_t* offs_by_orig;        // offsets by original index of set i
	// pool pass (single pass, no count/scan): records are bump-allocated in per-wave slabs from POOL_REGIONS + 1 regions of `records`
	unsigned long long* pool_cursor;         // cursor of region r at pool_cursor[r * POOL_CURSOR_STRIDE]: ints ASKED FOR so far by the waves of XCD r (r == POOL_OVERFLOW: by
	                                         // waves whose own region was full); [+ POOL_HITS_WORD] neighbour indices emitted, [+ POOL_WASTE_WORD] slab ints left unused
	const unsigned long long* pool_regions;  // device table: {first int, capacity in ints} of region 0..POOL_OVERFLOW (all capacities 0: a dry pass that only counts)
	uint32_t pool_slab;                      // ints a wave takes from a cursor per atomic (fast tier: its XCD's region; fat and general tier: region POOL_OVERFLOW)
	uint32_t pool_slab_heavy;                // the same for the fat and the general tier (their records are several hundred ints long: slabs of the first tier's
	                                         // size for a small set would be half empty)
	uint32_t* tickets;                 // ticket counters of the fast kernel: tickets[(xcd * CTRL_SUBRANGES + piece) * CTRL_STRIDE_U32] (zeroed before the launch)
	uint2* heavy;                      // worklist {first sorted position, key} of the cells the fast kernel skipped
	uint32_t* n_heavy;                 // its length (zeroed before the launch)
	uint32_t* tickets2; uint2* heavy2; uint32_t* n_heavy2;   // the same for the second tier (fat kernel -> general kernel)
	// sparse grid (blk_j != nullptr): no tables; the cells of set j / set i are found in their key-ordered occupied-cell lists through the block indices
	const uint2* socc_i; const uint32_t* blk_i; const uint2* socc_j; const uint32_t* blk_j; int sparse_shift;
	const uint32_t* abort_flag;   // the run's guard word (or nullptr): non-zero = the build already knows that this attempt will be thrown away (a point outside the
	                              // reused grid, an overflowed window of the one-read bucket pass -- the sorted arrays then have HOLES): the query kernels do nothing
	uint2* heavy0; uint32_t* n_heavy0;   // group formulation (tools/ubench/tnsx_query_group.hip, variant builds only): worklist of the cells it passes on to the three cell tiers (zeroed before)
};
// Control block of one pool pass.  Every hot counter sits CTRL_STRIDE_U32 words (4352 B) from the next: the L2 serialises
// atomics that hit the same cache line (measured: ~88 atomics/us per line, whatever the word), and the stride also spreads
// the counters over different L2 channels whether these interleave at 256 B or at 4 KiB.
static constexpr size_t CTRL_STRIDE_U32 = 1088;
#ifndef TNSX_CTRL_SUBRANGES
#define TNSX_CTRL_SUBRANGES 8
#endif
static constexpr uint32_t CTRL_SUBRANGES = TNSX_CTRL_SUBRANGES;   // ticket counters per XCD and tier (each hands out one contiguous piece of the XCD's cells)
static constexpr int POOL_REGIONS = 8, POOL_OVERFLOW = POOL_REGIONS;   // one region per XCD (fast tier) + the common region (heavy tiers, overflow)
static constexpr size_t POOL_CURSOR_STRIDE = CTRL_STRIDE_U32 / 2;      // in 64-bit words
static constexpr int POOL_HITS_WORD = 16, POOL_WASTE_WORD = 17;        // 64-bit words of a cursor's slot, on the line after the cursor's
static constexpr int POOL_CTRL_WORDS = 18;                             // what the host reads back per region
enum { CTRL_CURSOR = 0 /* POOL_REGIONS + 1 slots: u64 cursor | u64 hits, u64 waste */, CTRL_REGIONS = CTRL_CURSOR + POOL_REGIONS + 1 /* the region table */,
       CTRL_TICKETS = CTRL_REGIONS + 1 /* 8 x CTRL_SUBRANGES slots */, CTRL_NHEAVY = CTRL_TICKETS + 8 * CTRL_SUBRANGES,
       CTRL_TICKETS2 = CTRL_NHEAVY + 1 /* 8 x CTRL_SUBRANGES slots */, CTRL_NHEAVY2 = CTRL_TICKETS2 + 8 * CTRL_SUBRANGES,
       CTRL_NFILTERED = CTRL_NHEAVY2 + 1 /* length of the candidate-presence worklist */, CTRL_SLOTS = CTRL_NFILTERED + 1 };
static constexpr size_t CTRL_BYTES = CTRL_SLOTS * CTRL_STRIDE_U32 * sizeof(uint32_t);
enum { QUERY_COUNT = 0, QUERY_FILL = 1, QUERY_POOL = 2 };
struct QueryConfig {
	int arith;       // 0 strict, 1 contracted
	bool variable;   // per-point radii
	bool symmetric;  // d2 <= r_i^2 || d2 <= r_j^2 (only meaningful with variable)
	bool self;       // set_i == set_j: exclude the point itself
	int mode;        // QUERY_COUNT / QUERY_FILL (exact two-pass layout) / QUERY_POOL (single pass)
	bool groups = false;   // QUERY_POOL with a fixed radius: the group formulation (tools/ubench/tnsx_query_group.hip, variant builds only) instead of the three cell tiers
	int group_waves_per_cu = 0;   // its launch width (waves per CU); 0 = default
	int blocks_per_cu = 0, fast_blocks_per_cu = 0;   // launch widths (workgroups per CU) of the general / the fast kernels; 0 = default
	int tiers = 3;   // QUERY_POOL: bit 0 = the first tier, bit 1 = the two heavy tiers over the first tier's reject list (launched later, or not at all, when
	                 // the previous run of the pair rejected nothing: two empty launches are ~10 us of a step)
};
void launch_query(const QueryArgs& a, const QueryConfig& c, int n_compute_units, hipStream_t s);
// group formulation of a pool pass with a fixed radius (tools/ubench/tnsx_query_group.hip, variant builds only): k_query_groups over the occupied cells, then the three cell tiers
// over what it passed on (a.heavy0 / a.n_heavy0: at most one entry per occupied cell)
void launch_query_groups(const QueryArgs& a, const QueryConfig& c, int n_compute_units, hipStream_t s);
// pool pass over two different sets, candidate-presence filter: launch_mark_cells writes `value` into the byte of every grid cell
// that has an occupied cell of set j among its 27 (value 1 before the filter, 0 afterwards: the map is all zero between uses);
// launch_filter_marked compacts the occupied cells of set i whose byte is set into out / *n_out (zeroed before)
void launch_mark_cells(const uint2* occ_j, const uint32_t* n_occ_j, GridParams g, unsigned char* map, unsigned char value, size_t max_cells_j, hipStream_t s);
void launch_filter_marked(const uint2* occ_i, const uint32_t* n_occ_i, const unsigned char* map, uint2* out, uint32_t* n_out, size_t max_cells, hipStream_t s);

// ---- multi-GPU slab support (tnsx_kernels.hip) ----------------------------------------------------------------------
// ghost-halo selection of a slab decomposition along x; counts[2] must be zeroed before
void launch_halo_pack(const float* xyz, const float* radii, const long long* gids, int n, float left_cut, float right_cut, float* out_left,
                      float* out_right, unsigned long long cap_left, unsigned long long cap_right, unsigned int* counts, hipStream_t s);
// hist[clamp(trunc((x - x0) * inv_dx), 0, n_bins - 1)] += 1 for every point (hist is NOT zeroed here)
void launch_x_histogram(const float* xyz, int n, float x0, float inv_dx, int n_bins, unsigned int* hist, hipStream_t s);
// list entries j of the records of the first n_query points -> id_map[j], in place
void launch_translate_records(int* records, const uint64_t* offs_by_orig, int n_query, const int* id_map, int n_cus, hipStream_t s);

// slab layer (tnsx_slab.cpp): rows of W floats [x, y, z, (r,) gid_lo, gid_hi] received from a neighbour -> the tail of a set's [owned | ghosts]
// arrays.  count == nullptr: all n_rows rows exist; else the first *count do and the others become NaN points (x = NaN: the engine ignores them).
// *flag is set when a global id does not fit the 32-bit indices of the lists.
void launch_slab_unpack(const float* rows, uint32_t n_rows, const unsigned int* count, int W, float* xyz, float* radii, int* ids, unsigned int* flag, hipStream_t s);
void launch_slab_ids(const long long* gids, int n, int* ids, unsigned int* flag, hipStream_t s);          // ids[i] = (int)gids[i]; *flag on overflow
void launch_slab_flag_gt(const float* v, int n, float limit, unsigned int* flag, hipStream_t s);         // *flag = 1 if any v[i] > limit
void launch_slab_x_range(const float* xyz, int n, float* minmax, hipStream_t s);                          // minmax[0] = min(.., x), minmax[1] = max(.., x) (NaN skipped)
// redistribution (tnsx_slab_redistribute_begin): slab d owns cuts[d] <= x < cuts[d + 1] (world <= 64).  count_only: counters[d] += the points of slab d;
// else rows [x, y, z, (r,) gid_lo, gid_hi] of W floats, the points of slab d behind row first[d] in the order counters[d] (zeroed before) hands out
void launch_slab_dest_rows(bool count_only, const float* xyz, const float* radii, const long long* gids, int n, const float* cuts, int world, unsigned int* counters,
                           const unsigned int* first, float* rows, int W, hipStream_t s);
void launch_slab_rows_to_points(const float* rows, size_t n_rows, int W, float* xyz, float* radii, long long* gids, hipStream_t s);

// ---- start of a pool pass: the hot words of the pass's control block `ctrl` (CTRL_SLOTS slots) are zeroed and
//      regions[2 * (POOL_REGIONS + 1)] = {first int, capacity} of every region -> the device table the query reads (slot CTRL_REGIONS);
//      n_shared_empty > 0 (a pair of two different sets): offs[0..n) = 0 and records[0] = 0, the shared empty record at int 0 of the pool.
//      (On its own only for the repeat of a pass and for runs with more pool passes than launch_run_begin takes.)
void launch_pool_begin(const unsigned long long* regions, uint32_t* ctrl, uint64_t* offs, size_t n_shared_empty, int* records, hipStream_t s);
// dst[i] = src[i * stride + stride / 2] for i < ceil(*n_src / stride), *n_dst = that number (the worklist of a count-only pass over a sample of the occupied cells)
void launch_sample_cells(const uint2* src, const uint32_t* n_src, uint32_t stride, uint2* dst, uint32_t* n_dst, size_t max_dst, hipStream_t s);
// offs[p] = 0 (the pool's empty record) for every p < n whose x is NaN ("no point": it entered no cell and no query wrote its offset); xyz: 3 floats per point
void launch_point_nan_offsets(const float* xyz, int n, uint64_t* offs, hipStream_t s);
// the same for the exact layout (NaN-x points: the tail of the sorted array): phase 0 before the count pass (counts[p] = 1 for a query, 0 otherwise), phase 1 after the
// fill pass (count word 0 + offset of such a query)
void launch_exact_nan(const float4* xyzi_sorted, const uint32_t* orig_sorted, int n, uint32_t query_limit, uint32_t* counts, const uint64_t* offs_sorted, int* records,
                      uint64_t* offs_by_orig, int phase, hipStream_t s);
// ---- start of a run, ONE launch (round 4; every kernel of a step costs ~5 us of dispatch whatever it does):
//      words[0..n_words) = 0 (guard flag, partial checksums), n_occ[si] = 0 for every set si whose bit is set in `sets` (si < 64),
//      zero[k][0..n_zero[k]) = 0 (the cursors of the one-read bucket pass),
//      clear[k]: the table entries of the first n cells of a set's previous occupied-cell list are zeroed (what launch_table_clear does),
//      pool[k]:  what launch_pool_begin does, for every pool pass of the run
static constexpr int RUN_BEGIN_MAX_SETS = 8, RUN_BEGIN_MAX_POOLS = 8;
struct RunBeginClear { const uint2* occ; uint2* table; uint32_t n; };
struct RunBeginPool { unsigned long long regions[2 * (POOL_REGIONS + 1)]; uint32_t* ctrl; uint64_t* offs; size_t n_shared_empty; int* records; };
struct RunBeginArgs {
	unsigned long long* words; size_t n_words; uint32_t* n_occ; unsigned long long sets;
	int n_clear; RunBeginClear clear[RUN_BEGIN_MAX_SETS];
	int n_zero; uint32_t* zero[RUN_BEGIN_MAX_SETS]; uint32_t n_zero_words[RUN_BEGIN_MAX_SETS];
	int n_pool; RunBeginPool pool[RUN_BEGIN_MAX_POOLS];
};
void launch_run_begin(const RunBeginArgs& a, hipStream_t s);

// ---- end of a run: what the host reads after its one synchronisation, written into pinned host memory by one kernel (h_* are host pointers
//      of hipHostMalloc'ed memory; d_count / h_count may be nullptr)
static constexpr int RUN_END_MAX_JOBS = 8;
struct RunEndJob { const uint32_t* ctrl_cursor; unsigned long long* h_ctrl; const uint32_t* d_count; uint32_t* h_count;
                   const uint32_t* d_heavy; uint32_t* h_heavy; /* length of the first tier's reject list (nullptr: not wanted) */ };
struct RunEndArgs { RunEndJob job[RUN_END_MAX_JOBS]; int n_jobs; const uint32_t* n_occ; uint32_t* h_nocc; int n_sets; const unsigned long long* words; unsigned long long* h_words; size_t n_words; };
void launch_run_end(const RunEndArgs& a, hipStream_t s);

// ---- ascending order inside every record of the first n_query points (tnsx_options.sorted_lists), in place
void launch_sort_records(int* records, const uint64_t* offs_by_orig, int n_query, int n_cus, hipStream_t s);

// ---- gap-free copy of a pair's records in point order (the host mirror): len[p] = count + 1; out[new_offs[p] ...] = the record of point p (new_offs: n + 1 entries)
// indices_only: the count words are left out (len[p] = count, the copy starts behind the count word): a standard CSR
void launch_record_lengths(const int* records, const uint64_t* offs_by_orig, int n, uint32_t* len, bool indices_only, hipStream_t s);
void launch_compact_records(const int* records, const uint64_t* offs_by_orig, const uint64_t* new_offs, int n, int* out, bool indices_only, hipStream_t s);

// ---- permutation of byte records: out[new] = in[perm[new]] ------------------------------------------
void launch_permute_bytes(const void* in, void* out, const int* new_to_old, int n, size_t rec_bytes, hipStream_t s);

}  // namespace tnsx
