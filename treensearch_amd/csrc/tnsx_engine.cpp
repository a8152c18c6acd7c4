// Host engine behind include/tnsx.h: set registry, world box, device arenas, stage orchestration on one HIP
// stream, optional pinned host mirror of the neighbour lists.  Mirrors the control flow of
// tns::TreeNSearch::run() (TreeNSearch.cpp:138-149: _set_up, _check, _clear_neighborlists, world box, build, query)
// but none of its data structures: the build is sort-based on a uniform grid and lives entirely in HBM.
//
// There is deliberately NO CPU fallback anywhere in this file.
#include "../../include/tnsx.h"
#include "tnsx_kernels.h"
#include "tnsx_multi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_create_error;

// ------------------------------------------------------------------------------------------------ buffers
struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
	~DevBuf() { if (p) (void)hipFree(p); }
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
	// grow-only; contents are NOT preserved
	hipError_t reserve(size_t bytes)
	{
		if (bytes <= cap && p) return hipSuccess;
		if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
		size_t want = bytes + bytes / 8 + 256;   // 12.5 % slack so that slowly growing problems do not realloc every step
		hipError_t e = hipMalloc(&p, want);
		if (e != hipSuccess) { want = bytes < 256 ? 256 : bytes; e = hipMalloc(&p, want); }
		if (e == hipSuccess) cap = want;
		return e;
	}
	template <typename T> T* as() const { return static_cast<T*>(p); }
};
struct PinnedBuf {
	void* p = nullptr;
	size_t cap = 0;
	~PinnedBuf() { if (p) (void)hipHostFree(p); }
	PinnedBuf() = default;
	PinnedBuf(const PinnedBuf&) = delete;
	PinnedBuf& operator=(const PinnedBuf&) = delete;
	PinnedBuf(PinnedBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
	hipError_t reserve(size_t bytes)
	{
		if (bytes <= cap && p) return hipSuccess;
		if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
		const size_t want = bytes + bytes / 8 + 256;
		const hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
		if (e == hipSuccess) cap = want;
		return e;
	}
	template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct PointSet {
	const void* user_xyz = nullptr;
	const void* user_radii = nullptr;
	int n = 0;
	int n_query = -1;              // tnsx_set_query_count: only points [0, n_query) get lists (-1: all)
	const int* user_ids = nullptr; // tnsx_set_point_ids: device array, ids[j] is what the lists hold instead of j
	DevBuf orig_sorted;            // original index by sorted position (only with user ids: the sorted points then carry the id)
	bool is_double = false;        // dtype / memory space of the coordinate array ...
	bool on_device = false;
	bool radii_double = false;     // ... and of the radii array (kept separately: a points-only resize may change the former only)
	bool radii_on_device = false;
	bool has_radii = false;
	// staging (host inputs and/or double inputs)
	DevBuf raw_xyz, raw_radii;       // uploaded user bytes (host inputs)
	DevBuf f32_xyz, f32_radii;       // converted floats (double inputs)
	const float* d_xyz = nullptr;    // what the kernels read this run
	const float* d_radii = nullptr;
	// search structures
	DevBuf xyzi[2], r2[2];             // ping-pong of the cell sort; [sorted_buf] holds the sorted points of this run
	DevBuf table, occ;
	DevBuf blk;                    // sparse grid: block index of the occupied-cell list
	int sorted_buf = 0;
	// The cell table is never memset per run (it may be gigabytes for a sparse domain): the entries a run sets are exactly the
	// keys of its occupied-cell list, and the next run clears those first.  0 = all zero, 1 = `table_dirty` entries of `occ`
	// are set, 2 = unknown (a run failed half way): full memset.
	int table_state = 0;
	uint32_t table_dirty = 0;
	// static-set cache: checksum / identity of the input of the last run, and whether the two last runs saw the same input
	bool chk_valid = false, predicted_static = false, chk_double = false;
	uint64_t chk_value = 0;
	const void* chk_xyz = nullptr; const void* chk_radii = nullptr;
	int chk_n = -1;
	uint32_t built_gen = 0;        // grid generation the sorted arrays / table were built for (0: none)
	// one-read bucket pass (tnsx_build.hip k_bucket_scatter): windows of the intermediate array written by the last build, cursors of this run
	DevBuf bk_win, bk_cur;
	uint32_t bk_gen = 0;           // grid generation the windows were written for (0: none)
	int bk_n = 0, bk_buckets = 0;  // ... and the size of the set / the number of buckets then
	bool bk_now = false, bk_used = false;   // this attempt: the bucket build runs / with the one-read pass
	// zsort
	std::vector<int> zsort_host;    // filled on demand (zsort_host_order)
	int zsort_n = 0;
	DevBuf zsort_dev;
	bool zsort_ready = false;
};

struct PairResult {
	bool valid = false;
	int n_i = 0;
	int n_query = 0;             // points of set i that have a list: min(n_i, tnsx_set_query_count)
	uint64_t n_records = 0;      // extent of `records` in ints (pool mode: including slab holes and the unused ends of the regions)
	uint64_t n_neighbors = 0;
	uint64_t need_hint = 0;      // 1 + ints of records the previous run produced: sizes the pool of the next one (0: nothing known, dry pass first)
	uint32_t pool_slab = 16384;
	// pool mode: `records` is cut into one region per XCD + the common overflow region (tnsx_query.hip, PoolState)
	static constexpr int NR = tnsx::POOL_REGIONS + 1;
	uint64_t region_payload[NR] = { 0 };   // ints of records the waves of XCD r produced in the previous run
	uint64_t region_asked[NR] = { 0 };     // ints they asked their region for (records + the unused ends of their slabs); 0: not known for the current slab size
	uint64_t region_base[NR] = { 0 }, region_cap[NR] = { 0 }, region_used[NR] = { 0 };
	bool pooled = false;         // pool layout: int 0 of `records` is THE empty record (count 0), the regions start behind it
	bool shared_empty = false;   // ... and every offset is pre-set to it, so that cells without a candidate write nothing (pairs of two different sets)
	bool dry = false;            // this pass only counts (first run of a pair: nothing is known about its size yet)
	uint32_t sample_stride = 1;  // ... over every sample_stride-th occupied cell only (large sets: the counts are scaled up; the sized pass behind it is repaired if they fell short)
	uint32_t n_cells_i = 0;      // occupied cells of set i in the previous run
	bool groups_off = false;     // the group formulation sent too much of this pair to its leftover kernel: cell kernels from now on
	bool groups_now = false;     // this attempt runs the group formulation
	bool heavy_known = false;    // heavy_cells is what the previous run of this pair saw
	uint32_t heavy_cells = 0;    // cells its first tier passed on to the heavy tiers
	bool heavy_skipped = false;  // this attempt did not launch the heavy tiers (the previous run had nothing for them; checked after the run)
	DevBuf counts, offs_sorted, offs_orig, records, heavy, heavy2, filtered;
	PinnedBuf h_offs, h_records;
	bool mirrored = false;
};

}  // namespace

struct tnsx_context {
	tnsx_multi::State* multi = nullptr;   // multi-device mode (tnsx_options.n_devices > 1): everything is dispatched to tnsx_multi.cpp
	tnsx_options opt{};
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	int n_cus = 256;
	std::string last_error;

	std::vector<PointSet> sets;
	std::vector<std::vector<char>> active;      // active[i][j]
	std::vector<PairResult> pairs;              // [i * n_sets + j]
	int n_sets_at_last_run = 0;

	// radius / grid configuration (TreeNSearch.h:384-402)
	bool symmetric = true;
	bool radius_set = false;
	float radius = -1.0f, radius_sq = -1.0f;
	int n_sets_with_radii = 0;
	float cell_size = -1.0f, cell_size_inv = -1.0f;
	float world[6] = { FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX };   // bottom, top (octree_internals.h:29-30)
	int world_cells_pow2 = 0;
	bool scalar_world_box = false;   // this run is run_scalar(): the world box follows _update_world_AABB (no origin, see update_world_box)
	bool ran = false;
	bool cells_valid = false;   // the reference's are_cells_valid: a run() has happened since the sets last changed (selects the z-sort resolution)
	// the search grid of the last run and what it was laid out for (temporal reuse, see run_once)
	tnsx::GridParams grid{};
	float grid_h = 0.0f, grid_lo[3] = { 0, 0, 0 }, grid_hi[3] = { 0, 0, 0 }, grid_r_max = 0.0f;
	bool grid_valid = false, grid_variable = false;
	bool grid_box_scalar = false;   // the world box the grid was laid out under came from run_scalar()'s rule (no origin) / run()'s (united with the origin)
	bool grid_trimmed = false;      // the grid covers the bulk of the points only (trim_box)
	bool grid_sparse = false;       // no dense cell table: key-ordered lists of occupied cells + block indices (tnsx_build.hip, "SPARSE grids")
	int sparse_shift = 0; uint32_t sparse_blocks = 0;
	uint32_t grid_gen = 0;
	float zsort_inv_h = 0.0f;   // 1 / quantisation step of the last prepare_zsort (tnsx_stats.zsort_cell_size_inv)
	bool auto_dense_cells = true;
#ifdef TNSX_BUILD_NOSTORE
	static constexpr bool debug_nostore = true;    // a tools/ build for timing experiments only (tools/build_variant.sh -DTNSX_BUILD_NOSTORE): pool pass without its stores
#else
	static constexpr bool debug_nostore = false;   // the product: nothing at run time can make a pass skip its stores
#endif

	// scratch
	DevBuf bounds_partials, bounds_out, sort_temp, scan_temp, n_occ, permute_tmp, pool_ctrl, run_words, cell_map, trim_hist;
	DevBuf m_len, m_offs, m_records;   // staging of the host mirror's gap-free copy in point order (mirror_pair; shared by all pairs, used in stream order)
	PinnedBuf h_small, h_trim;
	tnsx_stats stats{};
	std::vector<hipEvent_t> events;
	std::mutex mirror_mutex;
	double sync_timeout_s = 0.0;   // slab layer watchdog: > 0 bounds the waits of a run on the stream (tnsx_internal_set_sync_timeout)
};

namespace {

#define TNSX_FAIL(ctx, code, ...)                                   \
	do {                                                            \
		char _b[512];                                               \
		std::snprintf(_b, sizeof(_b), __VA_ARGS__);                 \
		(ctx)->last_error = _b;                                     \
		return (code);                                              \
	} while (0)

#define HIPCHK(ctx, expr)                                                                               \
	do {                                                                                                \
		const hipError_t _e = (expr);                                                                   \
		if (_e != hipSuccess) TNSX_FAIL(ctx, TNSX_ERR_HIP, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
	} while (0)

inline bool set_ok(const tnsx_context* c, int s) { return s >= 0 && s < (int)c->sets.size(); }

void new_point_set(tnsx_context* c)
{
	// TreeNSearch.cpp:346-365: grow the active table with `false`
	c->sets.emplace_back();
	const size_t n = c->sets.size();
	for (auto& row : c->active) row.push_back(0);
	c->active.emplace_back(n, 0);
}

// world box update, TreeNSearch.cpp:474-521 (shared by the scalar and SIMD versions).
// simd_path: the tight bounds as run() and the no-tree prepare_zsort() see them (_update_world_AABB_simd, TreeNSearch.cpp:523-592):
// every thread of the reference takes the last points of its chunk as [x y z 0 0 0 0 0] (:564-569) and the final reduction
// folds lanes 3..5 into the result (:587-590), so that box is the tight box UNITED WITH THE ORIGIN whenever there is a point.
// run_scalar() (_update_world_AABB, :415-472) uses the tight box as it is.  Pinned against the reference's private
// domain_float by the `world` blocks of tests/golden/*.json.
tnsx_status update_world_box(tnsx_context* c, const float tight_in[6], bool simd_path)
{
	float tight[6];
	for (int d = 0; d < 6; d++) tight[d] = tight_in[d];
	if (simd_path) for (int d = 0; d < 3; d++) { tight[d] = std::min(tight[d], 0.0f); tight[3 + d] = std::max(tight[3 + d], 0.0f); }
	const float* wb = c->world; const float* wt = c->world + 3;
	if (wb[0] <= tight[0] && tight[3] <= wt[0] && wb[1] <= tight[1] && tight[4] <= wt[1] && wb[2] <= tight[2] && tight[5] <= wt[2]) {
		return TNSX_OK;
	}
	// computed into temporaries: the stored box changes only when the new one is legal (the reference exits here; this engine
	// returns an error and must stay consistent for the next call)
	float bottom[3], top[3], center[3];
	for (int d = 0; d < 3; d++) { bottom[d] = tight[d]; top[d] = tight[3 + d]; }
	for (int d = 0; d < 3; d++) center[d] = 0.5f * (top[d] + bottom[d]);
	float length = 0.0f;
	for (int d = 0; d < 3; d++) length = std::max(length, top[d] - bottom[d]);
	length += 100.0f * std::numeric_limits<float>::epsilon();
	length *= 1.1f;   // domain_enlargment, TreeNSearch.h:401
	const float cells_f = length / c->cell_size;
	if (!(cells_f < 32768.0f)) {   // (also catches inf / NaN; the cast below would be undefined for them)
		TNSX_FAIL(c, TNSX_ERR_GRID_TOO_LARGE, "TreeNSearch error: Max allowed cells per dimension is 32768 (2^15). Use set_cell_size() to set a larger value.");
	}
	const int n_cells = (int)cells_f + 1;
	int n_pow2 = 1;
	while (n_pow2 < n_cells) n_pow2 *= 2;
	length = c->cell_size * (float)n_pow2;
	c->world_cells_pow2 = n_pow2;
	for (int d = 0; d < 3; d++) {
		c->world[d] = center[d] - 0.5f * length;
		c->world[3 + d] = center[d] + 0.5f * length;
	}
	return TNSX_OK;
}

// stage user data of every set into device floats for this run
tnsx_status stage_inputs(tnsx_context* c)
{
	for (PointSet& s : c->sets) {
		s.d_xyz = nullptr; s.d_radii = nullptr;
		if (s.n == 0) continue;
		if (!s.user_xyz) TNSX_FAIL(c, TNSX_ERR_INVALID, "point set with n > 0 has a null coordinate pointer");
		const size_t esz = s.is_double ? sizeof(double) : sizeof(float);
		const size_t resz = s.radii_double ? sizeof(double) : sizeof(float);
		const void* dx = s.user_xyz;
		const void* dr = s.user_radii;
		if (s.has_radii && !s.user_radii) TNSX_FAIL(c, TNSX_ERR_INVALID, "variable-radius point set with n > 0 has a null radii pointer");
		if (!s.on_device) {
			HIPCHK(c, s.raw_xyz.reserve(3 * (size_t)s.n * esz));
			HIPCHK(c, hipMemcpyAsync(s.raw_xyz.p, s.user_xyz, 3 * (size_t)s.n * esz, hipMemcpyHostToDevice, c->stream));
			dx = s.raw_xyz.p;
		}
		if (s.has_radii && !s.radii_on_device) {
			HIPCHK(c, s.raw_radii.reserve((size_t)s.n * resz));
			HIPCHK(c, hipMemcpyAsync(s.raw_radii.p, s.user_radii, (size_t)s.n * resz, hipMemcpyHostToDevice, c->stream));
			dr = s.raw_radii.p;
		}
		if (s.is_double) {
			HIPCHK(c, s.f32_xyz.reserve(3 * (size_t)s.n * sizeof(float)));
			tnsx::launch_f64_to_f32((const double*)dx, s.f32_xyz.as<float>(), 3 * (size_t)s.n, c->stream);
			dx = s.f32_xyz.p;
		}
		if (s.has_radii && s.radii_double) {
			HIPCHK(c, s.f32_radii.reserve((size_t)s.n * sizeof(float)));
			tnsx::launch_f64_to_f32((const double*)dr, s.f32_radii.as<float>(), (size_t)s.n, c->stream);
			dr = s.f32_radii.p;
		}
		s.d_xyz = (const float*)dx;
		s.d_radii = s.has_radii ? (const float*)dr : nullptr;
	}
	return TNSX_OK;
}

// tight bounds + radius range over all sets -> host (one stream sync)
tnsx_status compute_bounds(tnsx_context* c, float out8[8])
{
	int total_partials = 0;
	for (const PointSet& s : c->sets) if (s.n > 0) total_partials += tnsx::bounds_num_blocks(s.n);
	out8[0] = out8[1] = out8[2] = FLT_MAX; out8[3] = out8[4] = out8[5] = -FLT_MAX; out8[6] = FLT_MAX; out8[7] = -FLT_MAX;
	if (total_partials == 0) return TNSX_OK;
	HIPCHK(c, c->bounds_partials.reserve((size_t)total_partials * 8 * sizeof(float)));
	HIPCHK(c, c->bounds_out.reserve(8 * sizeof(float)));
	HIPCHK(c, c->h_small.reserve(256));
	int off = 0;
	for (const PointSet& s : c->sets) {
		if (s.n == 0) continue;
		tnsx::launch_bounds_partial(s.d_xyz, s.d_radii, s.n, c->bounds_partials.as<float>() + 8 * (size_t)off, c->stream);
		off += tnsx::bounds_num_blocks(s.n);
	}
	tnsx::launch_bounds_final(c->bounds_partials.as<float>(), total_partials, c->bounds_out.as<float>(), c->stream);
	HIPCHK(c, hipMemcpyAsync(c->h_small.p, c->bounds_out.p, 8 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	std::memcpy(out8, c->h_small.p, 8 * sizeof(float));
	return TNSX_OK;
}

// A few points far away from all the others (a stray particle of an SPH scene) blow the bounding box up until no table of cells of
// one search radius fits any more; coarser cells would make EVERY query test thousands of candidates.  The grid does not have to
// cover every point, though: the binning clamps cell coordinates to the grid (bin_coord), clamping is monotone and never increases a
// coordinate difference, so two points at most one cell edge apart along an axis still land in the same or in adjacent cells and
// the distance test does the rest -- a grid over the BULK of the points is exact, the outliers just sit in its border cells.
// This finds that bulk: per-axis histograms of all points over the current box, tails of at most n/4096 points cut off on either
// side, repeated on the cut box (every round zooms in by up to the number of bins) until a table of cells of edge h0 fits.
// lo/hi: in = the tight bounds, out = the trimmed box.  Returns true if a box that fits was found.
static bool trim_box(tnsx_context* c, int64_t n_total, double h0, double margin, uint64_t cell_cap, float lo[3], float hi[3], tnsx_status* status)
{
	constexpr int NB = 2048;
	*status = TNSX_OK;
	auto fits = [&](const float* a, const float* b) {
		double cells = 1.0;
		for (int d = 0; d < 3; d++) cells *= std::floor(((double)b[d] - a[d] + 2.0 * margin) / h0) + 1.0;
		return cells <= (double)cell_cap;
	};
	if (c->trim_hist.reserve(3 * NB * sizeof(unsigned int)) != hipSuccess || c->h_trim.reserve(3 * NB * sizeof(unsigned int)) != hipSuccess) { *status = TNSX_ERR_HIP; return false; }
	const uint64_t budget = std::max<uint64_t>(1, (uint64_t)n_total >> 12);
	for (int round = 0; round < 6; round++) {
		if (fits(lo, hi)) return true;
		float inv[3];
		for (int d = 0; d < 3; d++) inv[d] = hi[d] > lo[d] ? (float)((double)NB / ((double)hi[d] - lo[d])) : 0.0f;
		if (hipMemsetAsync(c->trim_hist.p, 0, 3 * NB * sizeof(unsigned int), c->stream) != hipSuccess) { *status = TNSX_ERR_HIP; return false; }
		for (const PointSet& s : c->sets) {
			if (s.n == 0) continue;
			for (int d = 0; d < 3; d++) tnsx::launch_x_histogram(s.d_xyz + d, s.n, lo[d], inv[d], NB, c->trim_hist.as<unsigned int>() + d * NB, c->stream);
		}
		if (hipMemcpyAsync(c->h_trim.p, c->trim_hist.p, 3 * NB * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
		    hipStreamSynchronize(c->stream) != hipSuccess) { *status = TNSX_ERR_HIP; return false; }
		const unsigned int* h = c->h_trim.as<unsigned int>();
		bool progress = false;
		for (int d = 0; d < 3; d++) {
			if (!(hi[d] > lo[d])) continue;
			const unsigned int* hd = h + d * NB;
			int b0 = 0, b1 = NB - 1;
			uint64_t acc = 0;
			while (b0 < NB - 1 && acc + hd[b0] <= budget) acc += hd[b0++];
			acc = 0;
			while (b1 > b0 && acc + hd[b1] <= budget) acc += hd[b1--];
			const double dx = ((double)hi[d] - lo[d]) / NB;
			// one bin of slack on either side for the rounding of the binning
			const double nlo = std::max((double)lo[d], (double)lo[d] + (b0 - 1) * dx), nhi = std::min((double)hi[d], (double)lo[d] + (b1 + 2) * dx);
			if (nhi - nlo < 0.75 * ((double)hi[d] - lo[d])) progress = true;
			lo[d] = (float)nlo; hi[d] = (float)nhi;
		}
		if (!progress) break;
	}
	return fits(lo, hi);
}

// _set_up default cell size (TreeNSearch.cpp:300-316) and _check (TreeNSearch.cpp:366-392)
tnsx_status setup_and_check(tnsx_context* c, const float bounds8[8])
{
	if (c->cell_size < 0.0f) {
		float cs;
		if (c->radius_set) cs = 1.5f * c->radius;
		else {
			float min_radius = bounds8[6];
			if (min_radius == FLT_MAX) min_radius = 1.0f;
			cs = 1.5f * min_radius;
		}
		c->cell_size = cs;
		c->cell_size_inv = 1.0f / cs;
	}
	if (c->cell_size <= 0.0f) TNSX_FAIL(c, TNSX_ERR_CONFIG, "TreeNSearch error: cell_size is not set. Use TreeNSearch::set_cell_size().");
	if (c->radius_set && c->radius <= 0.0f) TNSX_FAIL(c, TNSX_ERR_CONFIG, "TreeNSearch error: global_search_radius <= 0.");
	if (c->radius_set && c->n_sets_with_radii > 0) TNSX_FAIL(c, TNSX_ERR_CONFIG, "TreeNSearch error: global search radius and per-point variable search radii specified.");
	if (!c->radius_set && c->n_sets_with_radii != (int)c->sets.size()) TNSX_FAIL(c, TNSX_ERR_CONFIG, "TreeNSearch error: not all point sets have per-point search radius specified.");
	return TNSX_OK;
}

int ceil_log2_u64(uint64_t v)
{
	int b = 0;
	while (((uint64_t)1 << b) < v) b++;
	return b;
}

struct StageTimer {
	tnsx_context* c;
	size_t next = 0;
	explicit StageTimer(tnsx_context* ctx) : c(ctx) {}
	int mark()
	{
		if (!c->opt.collect_stage_times) return -1;
		if (next >= c->events.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return -1; c->events.push_back(e); }
		(void)hipEventRecord(c->events[next], c->stream);
		return (int)next++;
	}
	float ms(int a, int b)
	{
		if (a < 0 || b < 0) return 0.f;
		float t = 0.f;
		(void)hipEventElapsedTime(&t, c->events[a], c->events[b]);
		return t;
	}
};

}  // namespace

// hipStreamSynchronize, or -- with a deadline (the slab layer's watchdog: an exchange that never completes must not hang the process) -- a poll
static tnsx_status sync_stream(tnsx_context* c)
{
	if (!(c->sync_timeout_s > 0.0)) { HIPCHK(c, hipStreamSynchronize(c->stream)); return TNSX_OK; }
	const auto t0 = std::chrono::steady_clock::now();
	for (;;) {
		const hipError_t q = hipStreamQuery(c->stream);
		if (q == hipSuccess) return TNSX_OK;
		if (q != hipErrorNotReady) TNSX_FAIL(c, TNSX_ERR_HIP, "HIP error %s while waiting for the stream", hipGetErrorString(q));
		if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->sync_timeout_s)
			TNSX_FAIL(c, TNSX_ERR_TIMEOUT, "the stream did not drain within %.1f s (watchdog)", c->sync_timeout_s);
		std::this_thread::yield();
	}
}

// ================================================================================================== C ABI
extern "C" {

int tnsx_version(void) { return TNSX_VERSION; }
int tnsx_query_formulation_available(int f)
{
#ifdef TNSX_WITH_GROUP_FORMULATION
	if (f == 1) return 1;
#endif
	return f == 0 ? 1 : 0;
}

tnsx_status tnsx_default_options(tnsx_options* opt)
{
	if (!opt) return TNSX_ERR_INVALID;
	std::memset(opt, 0, sizeof(*opt));
	opt->device_id = -1;
	opt->stream = nullptr;
	opt->arith = TNSX_ARITH_STRICT;
	opt->mirror_to_host = 0;
	opt->collect_stage_times = 0;
	opt->exact_layout = 0;
	opt->max_dense_cells = 0;
	opt->temporal_reuse = 1;
	opt->sorted_lists = 0;
	opt->n_devices = 0;
	return TNSX_OK;
}

const char* tnsx_last_error(const tnsx_context* ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

tnsx_status tnsx_create(const tnsx_options* opt, tnsx_context** out)
{
	if (!out) return TNSX_ERR_INVALID;
	*out = nullptr;
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
		g_create_error = "tnsx_create: no HIP device available (this engine has no CPU fallback)";
		return TNSX_ERR_NO_DEVICE;
	}
	tnsx_context* c = new tnsx_context();
	if (opt) c->opt = *opt; else tnsx_default_options(&c->opt);
	if (c->opt.n_devices > 1) {
		c->multi = tnsx_multi::create(c->opt, g_create_error);
		if (!c->multi) { delete c; return g_create_error.find("no HIP device") != std::string::npos ? TNSX_ERR_NO_DEVICE : TNSX_ERR_HIP; }
		*out = c;
		return TNSX_OK;
	}
	c->auto_dense_cells = c->opt.max_dense_cells == 0;   // default: bounded by the number of points (see run_once)
	if (c->opt.max_dense_cells == 0) c->opt.max_dense_cells = (uint64_t)1 << 30;
	if (c->opt.max_dense_cells > ((uint64_t)1 << 30)) c->opt.max_dense_cells = (uint64_t)1 << 30;   // 32-bit keys, int cell arithmetic
	int dev = c->opt.device_id;
	if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
	if (dev >= n_dev) { g_create_error = "tnsx_create: device_id out of range"; delete c; return TNSX_ERR_NO_DEVICE; }
	if (hipSetDevice(dev) != hipSuccess) { g_create_error = "tnsx_create: hipSetDevice failed"; delete c; return TNSX_ERR_HIP; }
	c->device = dev;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
		c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
		if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
			g_create_error = std::string("tnsx_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
			delete c;
			return TNSX_ERR_NO_DEVICE;
		}
	}
	if (c->opt.stream) { c->stream = (hipStream_t)c->opt.stream; c->own_stream = false; }
	else {
		if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_create_error = "tnsx_create: hipStreamCreate failed"; delete c; return TNSX_ERR_HIP; }
		c->own_stream = true;
	}
	*out = c;
	return TNSX_OK;
}

void tnsx_destroy(tnsx_context* c)
{
	if (!c) return;
	if (c->multi) { tnsx_multi::destroy(c->multi); delete c; return; }
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
	if (c->own_stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

// ------------------------------------------------------------------------------------------------ sets
#define TNSX_MULTI(call) do { if (c->multi) return (call); } while (0)
#define TNSX_NOT_MULTI(what) do { if (c->multi) TNSX_FAIL(c, TNSX_ERR_STATE, what ": not available on a multi-device context (host-resident inputs and host views only)"); } while (0)

int tnsx_add_point_set(tnsx_context* c, const void* xyz, const void* radii, int n, unsigned flags)
{
	if (!c) return -TNSX_ERR_INVALID;
	TNSX_MULTI(tnsx_multi::add_point_set(c->multi, xyz, radii, n, flags, c->last_error));
	if (n < 0) { c->last_error = "add_point_set: n_points < 0"; return -TNSX_ERR_INVALID; }
	new_point_set(c);
	c->cells_valid = false;   // TreeNSearch.cpp:364
	PointSet& s = c->sets.back();
	s.user_xyz = xyz; s.user_radii = radii; s.n = n;
	s.is_double = s.radii_double = (flags & TNSX_F64) != 0;
	s.on_device = s.radii_on_device = (flags & TNSX_DEVICE) != 0;
	s.has_radii = radii != nullptr;
	// a set declared through the radii overload is a variable-radius set even when n == 0 and radii == nullptr
	// (tests.cpp:453 hands null pointers for empty sets); the caller flags that case with TNSX_VARIABLE.
	if (flags & TNSX_VARIABLE) s.has_radii = true;
	if (s.has_radii) c->n_sets_with_radii++;
	return (int)c->sets.size() - 1;
}

tnsx_status tnsx_resize_point_set(tnsx_context* c, int set_id, const void* xyz, const void* radii, int n, unsigned flags)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_MULTI(tnsx_multi::resize_point_set(c->multi, set_id, xyz, radii, n, flags, c->last_error));
	if (!set_ok(c, set_id)) TNSX_FAIL(c, TNSX_ERR_INVALID, "TreeNSearch::resize_point_set error: Cannot resize a set that was not previously added.");
	if (n < 0) TNSX_FAIL(c, TNSX_ERR_INVALID, "resize_point_set: n_points < 0");
	PointSet& s = c->sets[set_id];
	const bool with_radii = radii != nullptr || (flags & TNSX_VARIABLE);
	if (with_radii && c->n_sets_with_radii == 0) {
		TNSX_FAIL(c, TNSX_ERR_INVALID, "TreeNSearch::resize_point_set error: Cannot resize a set with a radii array if it previously didn't have one.");
	}
	s.user_xyz = xyz; s.n = n;
	s.is_double = (flags & TNSX_F64) != 0;
	s.on_device = (flags & TNSX_DEVICE) != 0;
	if (with_radii) {
		// the flags describe the arrays handed over in THIS call; a points-only resize keeps the stored radii pointer together with
		// the dtype / memory space it was registered with (the reference keeps set_radii and set_radii_double apart, TreeNSearch.h:379-383)
		s.user_radii = radii;
		s.radii_double = s.is_double;
		s.radii_on_device = s.on_device;
	}
	s.zsort_ready = false;
	c->cells_valid = false;   // TreeNSearch.cpp:117-118
	return TNSX_OK;
}

// ------------------------------------------------------------------------------------------------ config
tnsx_status tnsx_set_search_radius(tnsx_context* c, float r)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_MULTI(tnsx_multi::set_search_radius(c->multi, r, c->last_error));
	if (c->n_sets_with_radii > 0) {
		TNSX_FAIL(c, TNSX_ERR_INVALID, "tns::TreeNSearch::set_search_radius error: Cannot set a global search radius if a set with a radii array was already added.");
	}
	c->radius_set = true;
	c->radius = r;
	c->radius_sq = r * r;   // TreeNSearch.cpp:29, fp32
	return TNSX_OK;
}
tnsx_status tnsx_set_cell_size(tnsx_context* c, float cell_size)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_MULTI(tnsx_multi::set_cell_size(c->multi, cell_size, c->last_error));
	if (c->cell_size > 0.0f) {
		TNSX_FAIL(c, TNSX_ERR_INVALID, "tns::TreeNSearch::set_cell_size error: Cell size already set. Create a new TreeNSearch instance if you need a different cell_size.");
	}
	c->cell_size = cell_size;
	c->cell_size_inv = 1.0f / cell_size;
	c->cells_valid = false;   // TreeNSearch.cpp:181
	return TNSX_OK;
}
tnsx_status tnsx_set_symmetric_search(tnsx_context* c, int active)
{
	if (!c) return TNSX_ERR_INVALID;
	if (c->multi) { tnsx_multi::set_symmetric(c->multi, active != 0); return TNSX_OK; }
	c->symmetric = active != 0;
	return TNSX_OK;
}
tnsx_status tnsx_set_arithmetic(tnsx_context* c, int arith)
{
	if (!c) return TNSX_ERR_INVALID;
	if (arith != TNSX_ARITH_STRICT && arith != TNSX_ARITH_CONTRACTED) TNSX_FAIL(c, TNSX_ERR_INVALID, "set_arithmetic: unknown mode %d", arith);
	if (c->multi) tnsx_multi::set_arithmetic(c->multi, arith);
	c->opt.arith = arith;
	return TNSX_OK;
}
tnsx_status tnsx_set_collect_stage_times(tnsx_context* c, int on)
{
	if (!c) return TNSX_ERR_INVALID;
	c->opt.collect_stage_times = on != 0;   // (multi-device contexts collect no stage times)
	return TNSX_OK;
}
tnsx_status tnsx_set_active_search(tnsx_context* c, int i, int j, int active)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_MULTI(tnsx_multi::set_active(c->multi, i, j, active != 0, c->last_error));
	if (!set_ok(c, i) || !set_ok(c, j)) TNSX_FAIL(c, TNSX_ERR_INVALID, "set_active_search: set does not exist (%d, %d)", i, j);
	c->active[i][j] = active != 0;
	return TNSX_OK;
}
tnsx_status tnsx_set_active_search_all(tnsx_context* c, int i, int search_in_all, int be_found_by_all)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_MULTI(tnsx_multi::set_active_all(c->multi, i, search_in_all != 0, be_found_by_all != 0, c->last_error));
	if (!set_ok(c, i)) TNSX_FAIL(c, TNSX_ERR_INVALID, "set_active_search: set does not exist (%d)", i);
	// column first, then row (TreeNSearch.cpp:223-232)
	for (size_t j = 0; j < c->sets.size(); j++) c->active[j][i] = be_found_by_all != 0;
	for (size_t j = 0; j < c->sets.size(); j++) c->active[i][j] = search_in_all != 0;
	return TNSX_OK;
}
tnsx_status tnsx_set_all_searches(tnsx_context* c, int active)
{
	if (!c) return TNSX_ERR_INVALID;
	if (c->multi) { tnsx_multi::set_all_searches(c->multi, active != 0); return TNSX_OK; }
	for (auto& row : c->active) for (auto& v : row) v = active != 0;
	return TNSX_OK;
}

// ------------------------------------------------------------------------------------------------ getters
int tnsx_get_n_sets(const tnsx_context* c) { return !c ? 0 : (c->multi ? tnsx_multi::n_sets(c->multi) : (int)c->sets.size()); }
int tnsx_get_n_points_in_set(const tnsx_context* c, int s)
{
	if (c && c->multi) return tnsx_multi::n_points_in_set(c->multi, s);
	return (c && set_ok(c, s)) ? c->sets[s].n : -1;
}
int64_t tnsx_get_total_n_points(const tnsx_context* c)
{
	if (c && c->multi) return tnsx_multi::total_points(c->multi);
	int64_t t = 0;
	if (c) for (const PointSet& s : c->sets) t += s.n;
	return t;
}
int tnsx_is_search_active(const tnsx_context* c, int i, int j)
{
	if (c && c->multi) return tnsx_multi::is_active(c->multi, i, j) ? 1 : 0;
	return (c && set_ok(c, i) && set_ok(c, j)) ? (int)c->active[i][j] : 0;
}
int tnsx_does_set_exist(const tnsx_context* c, int s) { return (c && s < tnsx_get_n_sets(c)) ? 1 : 0; }   // TreeNSearch.cpp:215-218
uint64_t tnsx_get_neighborlist_n_bytes(const tnsx_context* c)
{
	if (c && c->multi) return tnsx_multi::neighborlist_bytes(c->multi);
	uint64_t b = 0;
	if (c) for (const PairResult& p : c->pairs) if (p.valid) b += p.n_records * sizeof(int);
	return b;
}

// ------------------------------------------------------------------------------------------------ run
enum Stage { ST_UPLOAD, ST_BOUNDS, ST_KEYS, ST_SORT, ST_GATHER, ST_CELLS, ST_COUNT, ST_SCAN, ST_FILL, ST_MIRROR, ST_SORT_LISTS, ST_N };

// One attempt of run().  `speculate`: the search grid of the previous run is laid over the points without looking at their
// bounds first (no bounds kernel, no host round trip before the build), and sets that did not change between the last two runs
// keep their sorted arrays and cell table.  Both assumptions are checked on the device while the run proceeds (BuildGuard) and
// read back with the record totals at the one synchronisation every run has anyway; *redo tells the caller that one of them
// was wrong -- the run is then repeated without speculation.  This is the reference's own temporal reuse (the world box
// persists while it contains the points, TreeNSearch.cpp:474-482; unchanged sets, :77-79) moved off the critical path.
// the used parts of a pair's records (the shared empty record + what every pool region handed out) -> dst, same layout
static tnsx_status copy_records(tnsx_context* c, const PairResult& pr, int* dst, hipMemcpyKind kind, hipStream_t st)
{
	const int* src = pr.records.as<int>();
	if (pr.pooled) HIPCHK(c, hipMemcpyAsync(dst, src, sizeof(int), kind, st));
	for (int r = 0; r < PairResult::NR; r++) {
		if (pr.region_used[r] == 0) continue;
		HIPCHK(c, hipMemcpyAsync(dst + pr.region_base[r], src + pr.region_base[r], pr.region_used[r] * sizeof(int), kind, st));
	}
	return TNSX_OK;
}

// The pinned host mirror of one pair (what get_neighborlist reads on the CPU side): records WITHOUT the holes of the pool and in POINT order, offsets to match.
// Round 5: the link to the host bounds the drop-in mode (47 of 53 ms at C2), and a tenth of what crossed it were the unused ends of the waves' slabs.  The pair's
// records are compacted on the device first (lengths -> scan -> copy: ~1 ms for 2.4 GB), then 2.45 GB cross instead of 2.75; a CPU loop over the points reads its
// lists front to back.  The device views (offsets_device / records_device) are what they were; the host offsets differ from the device offsets.
static tnsx_status mirror_pair(tnsx_context* c, PairResult& pr, hipStream_t st)
{
	const size_t nq = (size_t)std::max(pr.n_query, 0);
	const uint64_t total = pr.n_neighbors + (uint64_t)nq;   // ints of the gap-free copy
	HIPCHK(c, pr.h_offs.reserve((std::max<size_t>(nq, 1) + 1) * sizeof(uint64_t)));
	HIPCHK(c, pr.h_records.reserve(std::max<uint64_t>(total, 1) * sizeof(int)));
	if (nq > 0) {
		// ONE staging area per context, not per pair (round-5 advice: a second persistent copy of every pair's records doubled the list memory of the drop-in
		// mode): everything here is enqueued on the context's stream, so the copy of one pair to the host is over before the next pair's compaction overwrites it
		HIPCHK(c, c->m_len.reserve(nq * sizeof(uint32_t)));
		HIPCHK(c, c->m_offs.reserve((nq + 1) * sizeof(uint64_t)));
		HIPCHK(c, c->m_records.reserve(std::max<uint64_t>(total, 1) * sizeof(int)));
		HIPCHK(c, c->scan_temp.reserve(tnsx::scan_temp_bytes(nq)));
		tnsx::launch_record_lengths(pr.records.as<int>(), pr.offs_orig.as<uint64_t>(), (int)nq, c->m_len.as<uint32_t>(), false, st);
		tnsx::exclusive_scan_u32_to_u64(c->m_len.as<uint32_t>(), c->m_offs.as<uint64_t>(), nq, c->scan_temp.p, st);
		tnsx::launch_compact_records(pr.records.as<int>(), pr.offs_orig.as<uint64_t>(), c->m_offs.as<uint64_t>(), (int)nq, c->m_records.as<int>(), false, st);
		HIPCHK(c, hipMemcpyAsync(pr.h_offs.p, c->m_offs.p, nq * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
		if (total > 0) HIPCHK(c, hipMemcpyAsync(pr.h_records.p, c->m_records.p, total * sizeof(int), hipMemcpyDeviceToHost, st));
	}
	pr.mirrored = true;
	return TNSX_OK;
}

// slab size, regions and record storage of a pool pass.  payload[r]: ints of records region r is expected to receive (nullptr: dry
// pass); asked[r]: what the waves asked region r for in the previous pass with the same slab size -- records plus the unused ends
// of their slabs, which is what the region has to hold (nullptr after a dry pass, whose slabs have another size: the regions then
// get a quarter / a half more than the records need).  generous: the common region can take EVERYTHING (the redo of a pass that
// overflowed must not overflow again).
#ifndef TNSX_POOL_SLAB_DIV
#define TNSX_POOL_SLAB_DIV 8   // slabs of 1 / DIV of a wave's share of the pool (the holes: half of every wave's last slab)
#endif
static tnsx_status size_pool(tnsx_context* c, PairResult& pr, const uint64_t* payload, const uint64_t* asked, bool generous, int query_waves)
{
	for (int r = 0; r < PairResult::NR; r++) { pr.region_base[r] = 0; pr.region_cap[r] = 0; pr.region_used[r] = 0; }
	if (!payload) {
		pr.pool_slab = 16384;   // nothing is written, big slabs keep the cursor atomics rare
		HIPCHK(c, pr.records.reserve(1024 * sizeof(int)));
		return TNSX_OK;
	}
	uint64_t total = 0;
	for (int r = 0; r < PairResult::NR; r++) total += payload[r];
	const uint64_t expect = total + total / 8 + 1024;
	// a wave's last slab stays half empty on average: slabs of 1/8 of a wave's share keep the holes at ~6 % of the pool
	// (round 3: the unit of allocation is the block of a whole cell -- some hundred ints to a few thousand -- so a slab is at least 4096
	//  ints: with the 256-int slabs a small set used to get, every cell would be an allocation of its own)
	uint64_t slab = std::min<uint64_t>(16384, std::max<uint64_t>(4096, expect / ((uint64_t)query_waves * TNSX_POOL_SLAB_DIV)));
	// (the holes depend on the slab size: the previous size is kept while it is within an eighth of the ideal one)
	if (asked && slab >= (uint64_t)pr.pool_slab - pr.pool_slab / 8 && slab <= (uint64_t)pr.pool_slab + pr.pool_slab / 8) slab = pr.pool_slab;
	if (asked && slab != pr.pool_slab) asked = nullptr;
	pr.pool_slab = (uint32_t)slab;
	const uint64_t slab_heavy = std::max<uint64_t>(slab, 8192);
	// (a wave takes a slab only if it gets a cell: small sets keep small pools)
	const uint64_t waves_all = std::min<uint64_t>((uint64_t)query_waves, (uint64_t)pr.n_i + 8);
	const uint64_t waves_x = std::min<uint64_t>((uint64_t)query_waves / tnsx::POOL_REGIONS, (uint64_t)pr.n_i / tnsx::POOL_REGIONS + 2);
	// (a wave of the heavy tiers that gets a cell writes at least a handful of records; should this ever be too little, the pass is repeated)
	const uint64_t waves_heavy = std::min<uint64_t>(std::min<uint64_t>((uint64_t)query_waves, (uint64_t)pr.n_i / 4 + 8), payload[tnsx::POOL_OVERFLOW] / 16 + 8);
	uint64_t first = 64;   // (int 0: the empty record of the pool)
	for (int r = 0; r < PairResult::NR; r++) {
		const bool common = r == tnsx::POOL_OVERFLOW;
		uint64_t cap;
		// steady state: what was asked for last time + 6 % (the common region: + 6 % of everything, for what the others cannot hold);
		// after a dry pass: the records + a quarter (the common region: a half) + a slab per wave that can get a cell
		if (asked) {
			cap = asked[r] + asked[r] / 16 + 1024 + (common ? expect / 16 + 4096 : 0);
			// (the common region of a small dense set: how many waves of the heavy tiers get a cell -- and open a slab of their own, to leave
			//  it mostly empty -- is decided by the race for the tickets and moved `asked` by 17 % between two runs of a 10 000-point set;
			//  every wave that can get a cell may open one, up to twice what was asked for last time)
			if (common) cap += std::min<uint64_t>(std::min<uint64_t>((uint64_t)query_waves + (uint64_t)query_waves / 4, (uint64_t)std::max(pr.n_cells_i, 1u)),
			                                      std::max<uint64_t>(asked[r] / slab_heavy, 64)) * slab_heavy;
		}
		else cap = payload[r] + payload[r] / (common ? 2 : 4) + 1024 + (common ? expect / 16 + waves_heavy * slab_heavy : waves_x * slab);
		if (generous && common) cap += expect + waves_all * slab_heavy;
		pr.region_base[r] = first; pr.region_cap[r] = cap;
#ifdef TNSX_BUILD_DEBUG_POOL
		fprintf(stderr, "[tnsx] size_pool region %d: payload %llu asked %lld cap %llu slab %llu n_i %d cells_prev %u\n", r, (unsigned long long)payload[r], asked ? (long long)asked[r] : -1ll, (unsigned long long)cap, (unsigned long long)slab, pr.n_i, pr.n_cells_i);
#endif
		first = (first + cap + 63) & ~(uint64_t)63;
	}
	HIPCHK(c, pr.records.reserve(first * sizeof(int)));
	return TNSX_OK;
}

// A run that does not reuse the previous run's grid: world box of the reference semantics, then the search grid -- the box it covers (tight bounds widened by
// two radii, inside the world box), the cell edge (r_max with the rounding margin of the binning), dense table / trimmed to the bulk of the points / sparse /
// coarsened.  b8: tight bounds and radius range of all points (compute_bounds).  Leaves the result in c->grid* (grid_valid, grid_gen, ...).
static tnsx_status layout_grid(tnsx_context* c, const float b8[8], int64_t n_total, bool variable)
{
	tnsx::GridParams g{};
	// ---- world box of the reference semantics (kept for zsort + the 2^15 cells/dimension limit)
	if (n_total > 0) { const tnsx_status r = update_world_box(c, b8, !c->scalar_world_box); if (r != TNSX_OK) return r; }
	const float r_max = variable ? b8[7] : c->radius;
	if (n_total > 0 && !(r_max > 0.0f) ) TNSX_FAIL(c, TNSX_ERR_CONFIG, "TreeNSearch error: search radius must be > 0");
	if (n_total > 0 && !std::isfinite(r_max)) TNSX_FAIL(c, TNSX_ERR_INVALID, "a search radius is not finite");
	// ---- the box the grid is laid over: the tight bounds widened by two cell edges on every side, but never beyond the
	//      world box -- as long as every point stays inside it, the reference would keep its world box too
	//      (TreeNSearch.cpp:474-482), so a later run may reuse this grid AND the world box without seeing the bounds.
	// ---- search grid: cell edge h >= r_max with a margin that covers the fp32 rounding of the binning, so that any
	//      pair the fp32 predicate can accept lies in adjacent cells.  Coarsened until the dense table fits.
	c->grid_valid = false;
	if (n_total > 0) {
		// the dense table costs 8 bytes per cell and set: bounded by the number of points (a sparse scene trims the grid to the
		// bulk of its points or coarsens its cells instead of allocating gigabytes), and by the option
		const uint64_t cell_cap = c->auto_dense_cells ? std::min<uint64_t>(c->opt.max_dense_cells, std::max<uint64_t>((uint64_t)1 << 22, 64ull * (uint64_t)n_total))
		                                              : c->opt.max_dense_cells;
		float blo[3] = { b8[0], b8[1], b8[2] }, bhi[3] = { b8[3], b8[4], b8[5] };
		bool trimmed = false;
		{
			double cells = 1.0;
			const double h0 = (double)r_max * 1.001;
			for (int d = 0; d < 3; d++) cells *= std::floor(((double)bhi[d] - blo[d] + 4.0 * r_max) / h0) + 1.0;
			if (cells > (double)cell_cap) {
				// cells of one search radius over the bounding box do not fit: far outliers?  (trim_box)
				float tlo[3] = { blo[0], blo[1], blo[2] }, thi[3] = { bhi[0], bhi[1], bhi[2] };
				tnsx_status ts = TNSX_OK;
				if (trim_box(c, n_total, h0, 2.0 * r_max, cell_cap, tlo, thi, &ts)) {
					for (int d = 0; d < 3; d++) { blo[d] = tlo[d]; bhi[d] = thi[d]; }
					trimmed = true;
				}
				if (ts != TNSX_OK) TNSX_FAIL(c, ts, "HIP error while trimming the search grid to the bulk of the points");
			}
		}
		float lo[3], hi[3];
		for (int d = 0; d < 3; d++) {
			const float m = 2.0f * r_max;
			lo[d] = std::max(blo[d] - m, c->world[d]);
			hi[d] = std::min(bhi[d] + m, c->world[3 + d]);
			if (!(lo[d] <= blo[d])) lo[d] = blo[d];            // (a world box that does not contain the points: a failed update)
			if (!(hi[d] >= bhi[d])) hi[d] = bhi[d];
		}
		const double ext[3] = { (double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2] };
		const double max_ext = std::max(ext[0], std::max(ext[1], ext[2]));
		const double n0 = std::floor(max_ext / (double)r_max) + 2.0;
		double h = (double)r_max * (1.0 + 8.0 * 5.9604644775390625e-08 * (n0 + 2.0)) * (1.0 + 1e-6);
		bool fits = false;
		// A grid whose dense table does not fit keeps its cell edge as a SPARSE grid (round 4: lists of occupied cells instead of a table) as long as
		// its cells can be numbered with 32 bits; only beyond that are the cells coarsened (exact either way; coarser cells cost candidates).
		// (sparse_grid > 0: always; 0: when the bound of the dense table is the automatic one -- a caller who sets max_dense_cells asked for THAT many
		//  cells -- and the cell edge would have to double.  Measured, tools/sparse_probe.py, a 10 M-point filament through the whole box with 3 - 6 points
		//  per cell: cells 1.6 x coarser make the run 2.04 ms, the sparse grid at one radius 4.07 ms -- such clouds pay per CELL, the coarser grid has
		//  fewer, and the sparse grid is served by the general kernel alone.  The sparse grid wins where coarsening costs candidates by the cube.)
		const bool sparse_ok = c->opt.sparse_grid > 0 || (c->opt.sparse_grid == 0 && c->auto_dense_cells);
		auto cells_at = [&](double hh) { return (std::floor(ext[0] / hh) + 1.0) * (std::floor(ext[1] / hh) + 1.0) * (std::floor(ext[2] / hh) + 1.0); };
		const double cells0 = cells_at(h);
		const bool want_sparse = sparse_ok && cells0 > (double)cell_cap && (c->opt.sparse_grid > 0 || cells0 > 8.0 * (double)cell_cap);
		const double limit = want_sparse ? 4294967280.0 : (double)cell_cap;   // (a grid beyond 32-bit keys is coarsened until it fits as a sparse one)
		bool sparse = false;
		for (int it = 0; it < 400 && !fits; it++) {
			const double nx = std::floor(ext[0] / h) + 1.0, ny = std::floor(ext[1] / h) + 1.0, nz = std::floor(ext[2] / h) + 1.0;
			if (nx * ny * nz <= limit && nx < 2.0e6 && ny < 2.0e6 && nz < 2.0e6) {
				g.nx = (int)nx; g.ny = (int)ny; g.nz = (int)nz;
				sparse = nx * ny * nz > (double)cell_cap;
				fits = true;
			}
			else h *= 1.26;
		}
		c->grid_sparse = sparse;
		if (!fits) TNSX_FAIL(c, TNSX_ERR_INVALID, "no search grid fits the extent of the points");
		g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2];
		float hf = (float)h;
		if ((double)hf < h) hf = std::nextafter(hf, FLT_MAX);
		g.inv_h = 1.0f / hf;
		if ((double)g.inv_h * (double)hf > 1.0) g.inv_h = std::nextafter(g.inv_h, 0.0f);   // never overestimate 1/h
		c->grid = g;
		c->grid_h = hf;
		for (int d = 0; d < 3; d++) { c->grid_lo[d] = lo[d]; c->grid_hi[d] = hi[d]; }
		c->grid_r_max = r_max;
		c->grid_variable = variable;
		c->grid_box_scalar = c->scalar_world_box;
		c->grid_valid = true;
		c->grid_trimmed = trimmed;
		c->grid_gen++;
	}
	else { g.nx = g.ny = g.nz = 1; g.inv_h = 1.0f; c->grid = g; c->grid_h = 0.0f; c->grid_trimmed = false; c->grid_sparse = false; }
	return TNSX_OK;
}

// ==================================================================================================
// One attempt of run(), in stages (round 5: this was ONE function of 557 lines; the reference's run() is ten lines that call stages,
// TreeNSearch.cpp:138-149).  RunAttempt holds what the stages share -- nothing in it outlives the attempt:
//   plan_run         everything decided on the host before the first launch: inputs, bounds (unless the grid is reused), the grid, the result slots, the
//                    pool of every pair sized from the previous run's numbers
//   launch_build     k_run_begin + the build of every set that does not keep its structures
//   launch_queries   the pass of every active pair + k_run_end
//   judge_attempt    the ONE synchronisation; were the assumptions of the attempt right?
//   finish_run       per pair: the counters of its pass, a repair if a pool region overflowed (collect_pair); sorted lists, host mirror, statistics
// ==================================================================================================
namespace {
struct RunJob { int i, j; bool pool; bool begun; };
struct RunSpan { int stage, a, b; };
constexpr size_t RUN_WB = (size_t)tnsx::CHK_SLOTS * tnsx::CHK_STRIDE;                 // 64-bit words per block of the attempt's device words
constexpr size_t RUN_HC = (size_t)PairResult::NR * tnsx::POOL_CTRL_WORDS;             // per job: cursor / neighbours / unused ints of every pool region (exact layout: word 0 = total)
struct RunAttempt {
	tnsx_context* c;
	bool speculate;
	StageTimer tm;
	std::vector<RunSpan> spans;
	int n_sets = 0;
	hipStream_t st = nullptr;
	int64_t n_total = 0;
	bool variable = false, sparse = false;
	tnsx::GridParams g{};
	uint64_t n_cells = 1;
	int key_bits = 1;
	unsigned long long* d_words = nullptr;   // block 0 = the guard flag, block 1 + si = the partial checksums of set si
	std::vector<char> skipped;               // sets that keep their build in this attempt
	std::vector<RunJob> jobs;                // the active pairs
	uint64_t* h_ctrl = nullptr; uint64_t* h_words = nullptr;
	uint32_t* h_nocc = nullptr; uint32_t* h_filt = nullptr; uint32_t* h_left = nullptr; uint32_t* h_heavy = nullptr;
	tnsx::RunEndArgs run_end{};
	bool defer_readback = true;
	int query_waves = 0;
	tnsx::QueryConfig qc{};
	int e_begin = -1, t_build0 = -1;
	RunAttempt(tnsx_context* ctx, bool spec) : c(ctx), speculate(spec), tm(ctx) {}
	void span(int stage, int a, int b) { if (a >= 0 && b >= 0) spans.push_back({ stage, a, b }); }
	uint32_t* ctrl_slot(size_t k, int slot) const { return c->pool_ctrl.as<uint32_t>() + (k * tnsx::CTRL_SLOTS + (size_t)slot) * tnsx::CTRL_STRIDE_U32; }
	PairResult& pair(const RunJob& jb) const { return c->pairs[(size_t)jb.i * n_sets + jb.j]; }
	bool keeps_its_build(const PointSet& s) const
	{
		const bool same_input = s.chk_valid && s.chk_xyz == s.user_xyz && s.chk_radii == s.user_radii && s.chk_n == s.n && s.chk_double == s.is_double;
		const bool cacheable = !s.user_ids && s.n > 0;
		return speculate && cacheable && s.predicted_static && same_input && s.built_gen == c->grid_gen && (sparse || s.table_state == 1);
	}
};
// (the stages keep the names the one function had for what they share)
#define TNSX_RUN_ALIASES                                                                                                      \
	tnsx_context* const c = run.c; const bool speculate = run.speculate; StageTimer& tm = run.tm; const int n_sets = run.n_sets;      \
	hipStream_t st = run.st; tnsx_stats& S = c->stats; const int64_t n_total = run.n_total; const bool variable = run.variable;     \
	const bool sparse = run.sparse; const tnsx::GridParams g = run.g; const uint64_t n_cells = run.n_cells; const int key_bits = run.key_bits; \
	constexpr size_t WB = RUN_WB, HC = RUN_HC; unsigned long long* const d_words = run.d_words; std::vector<char>& skipped = run.skipped; \
	std::vector<RunJob>& jobs = run.jobs; uint64_t* const h_ctrl = run.h_ctrl; uint64_t* const h_words = run.h_words; uint32_t* const h_nocc = run.h_nocc; \
	uint32_t* const h_filt = run.h_filt; uint32_t* const h_left = run.h_left; uint32_t* const h_heavy = run.h_heavy; tnsx::RunEndArgs& run_end = run.run_end; \
	bool& defer_readback = run.defer_readback; const int query_waves = run.query_waves; tnsx::QueryConfig& qc = run.qc;          \
	auto span = [&run](int stage_, int a_, int b_) { run.span(stage_, a_, b_); };                                                 \
	auto ctrl_slot = [&run](size_t k_, int slot_) { return run.ctrl_slot(k_, slot_); };                                           \
	auto keeps_its_build = [&run](const PointSet& s_) { return run.keeps_its_build(s_); };                                        \
	(void)speculate; (void)tm; (void)n_sets; (void)st; (void)S; (void)n_total; (void)variable; (void)sparse; (void)g; (void)n_cells; (void)key_bits; (void)WB; (void)HC; \
	(void)d_words; (void)skipped; (void)jobs; (void)h_ctrl; (void)h_words; (void)h_nocc; (void)h_filt; (void)h_left; (void)h_heavy; (void)run_end; (void)defer_readback; \
	(void)query_waves; (void)qc; (void)span; (void)ctrl_slot; (void)keeps_its_build
}  // namespace

static tnsx_status plan_run(RunAttempt& run)
{
	tnsx_context* const c = run.c; const bool speculate = run.speculate; StageTimer& tm = run.tm;
	auto span = [&run](int stage_, int a_, int b_) { run.span(stage_, a_, b_); };
	constexpr size_t WB = RUN_WB, HC = RUN_HC;
	const int n_sets = run.n_sets = (int)c->sets.size();
	run.st = c->stream;
	tnsx_stats& S = c->stats;
	const int retries_so_far = S.speculation_redos;
	std::memset(&S, 0, sizeof(S));
	S.speculation_redos = retries_so_far;
	S.n_sets = n_sets;
	c->ran = false;

	const int e_begin = run.e_begin = tm.mark();
	// ---- inputs -> device floats
	{ const tnsx_status r = stage_inputs(c); if (r != TNSX_OK) return r; }
	const int e_up = tm.mark();
	span(ST_UPLOAD, e_begin, e_up);

	int64_t n_total = 0;
	for (const PointSet& s : c->sets) n_total += s.n;
	run.n_total = n_total;
	S.n_points = (uint64_t)n_total;
	const bool variable = run.variable = !c->radius_set;

	// ---- bounds (tight AABB, radius range) -> host, unless the previous run's grid is reused
	float b8[8] = { FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX, FLT_MAX, -FLT_MAX };
	if (!speculate) {
		const tnsx_status r = compute_bounds(c, b8);
		if (r != TNSX_OK) return r;
		for (int k = 0; k < 6; k++) {
			if (n_total > 0 && !std::isfinite(b8[k])) TNSX_FAIL(c, TNSX_ERR_INVALID, "a point coordinate is not finite (inf, or NaN in y / z)");
		}
	}
	const int e_bounds = tm.mark();
	span(ST_BOUNDS, e_up, e_bounds);
	{ const tnsx_status r = setup_and_check(c, b8); if (r != TNSX_OK) return r; }

	// ---- result slots (TreeNSearch.cpp:393-413)
	c->pairs.resize((size_t)n_sets * n_sets);
	for (PairResult& p : c->pairs) { p.valid = false; p.mirrored = false; }
	c->n_sets_at_last_run = n_sets;

	tnsx::GridParams g{};
	uint64_t n_cells = 1;
	if (!speculate) { const tnsx_status r = layout_grid(c, b8, n_total, variable); if (r != TNSX_OK) return r; }
	g = run.g = c->grid;
	n_cells = run.n_cells = (uint64_t)g.nx * g.ny * g.nz;
	const bool sparse = run.sparse = c->grid_sparse;
	S.grid_cell_size = c->grid_h;
	for (int d = 0; d < 3; d++) { S.world_bottom[d] = c->world[d]; S.world_top[d] = c->world[3 + d]; }
	S.world_cells_pow2 = c->world_cells_pow2;
	S.grid_dims[0] = g.nx; S.grid_dims[1] = g.ny; S.grid_dims[2] = g.nz;
	S.grid_origin[0] = g.ox; S.grid_origin[1] = g.oy; S.grid_origin[2] = g.oz;
	S.n_grid_cells = n_cells;
	S.grid_trimmed = c->grid_trimmed ? 1 : 0;
	S.grid_sparse = sparse ? 1 : 0;
	const int key_bits = run.key_bits = std::max(1, ceil_log2_u64(n_cells + 1));   // + 1: the key behind the last cell, where NaN points ("no point") go
	S.key_bits = key_bits;
	if (sparse) { c->sparse_shift = std::max(0, key_bits - 22); c->sparse_blocks = (uint32_t)(((n_cells - 1) >> c->sparse_shift) + 1); }
	S.radix_passes = tnsx::cell_sort_plan(key_bits).passes;
	S.speculated = speculate ? 1 : 0;

	// ---- device words of this attempt (64-bit each): block 0 = the guard flag, block 1 + si = the partial checksums of set si
	HIPCHK(c, c->run_words.reserve(sizeof(uint64_t) * WB * (size_t)(n_sets + 1)));
	run.d_words = c->run_words.as<unsigned long long>();

	// ---- per set: cell sort -> cell table (or nothing: a set that did not change keeps what it has)
	{
		const void* old = c->n_occ.p;
		HIPCHK(c, c->n_occ.reserve(sizeof(uint32_t) * (size_t)std::max(n_sets, 1)));
		if (c->n_occ.p != old) for (PointSet& s : c->sets) s.built_gen = 0;   // the occupied-cell counts of cached sets lived in the old buffer
	}
	run.skipped.assign((size_t)n_sets, 0);
	// ---- per active pair: what its pass needs on the host side (sizes from the previous run; nothing here waits for the device).
	//      pool mode (default): ONE pass, records bump-allocated from the regions of the pair's pool (first run of a pair: a dry pass first);
	//      exact mode (opt.exact_layout): count -> scan -> fill, gap-free CSR in sorted order.
	std::vector<RunJob>& jobs = run.jobs;
	for (int i = 0; i < n_sets; i++) for (int j = 0; j < n_sets; j++) if (c->active[i][j]) jobs.push_back({ i, j, false, false });
	HIPCHK(c, c->h_small.reserve(sizeof(uint64_t) * (HC * jobs.size() + 2 + WB * ((size_t)n_sets + 1)) + sizeof(uint32_t) * (size_t)(n_sets + 1 + 3 * (jobs.size() + 1)) + 64));
	uint64_t* h_ctrl = run.h_ctrl = c->h_small.as<uint64_t>();
	uint64_t* h_words = run.h_words = h_ctrl + HC * jobs.size() + 2;                  // guard flag, partial checksums
	uint32_t* h_nocc = run.h_nocc = reinterpret_cast<uint32_t*>(h_words + WB * ((size_t)n_sets + 1));
	uint32_t* h_filt = run.h_filt = h_nocc + n_sets + 1;                             // per job: cells that passed the candidate-presence filter
	for (size_t k = 0; k < jobs.size(); k++) h_filt[k] = 0;
	uint32_t* h_left = run.h_left = h_filt + jobs.size() + 1;                        // per job: cells the group kernel passed on to the cell tiers
	for (size_t k = 0; k < jobs.size(); k++) h_left[k] = 0;
	uint32_t* h_heavy = run.h_heavy = h_left + jobs.size() + 1;                       // per job: cells the first tier passed on to the heavy tiers
	for (size_t k = 0; k < jobs.size(); k++) h_heavy[k] = 0;
	// the words of the pool passes go to the host with the ONE kernel at the end of the attempt (launch_run_end); the repeat of a pass that
	// overflowed, and attempts with more pool passes than that kernel takes, copy them pass by pass
	run.run_end = tnsx::RunEndArgs{};
	run.defer_readback = true;
	HIPCHK(c, c->pool_ctrl.reserve(tnsx::CTRL_BYTES * (jobs.size() + 1)));   // per job: cursor, hit_total, 2 x (8 tickets, n_heavy), spread out
	const int query_waves = run.query_waves = c->n_cus * 8 * 4;
	for (size_t k = 0; k < jobs.size(); k++) {
		RunJob& jb = jobs[k];
		PairResult& pr = run.pair(jb);
		const int n_i = c->sets[jb.i].n;
		pr.n_i = n_i;
		pr.n_query = c->sets[jb.i].n_query < 0 ? n_i : std::min(n_i, c->sets[jb.i].n_query);
		HIPCHK(c, pr.offs_orig.reserve((size_t)std::max(n_i, 1) * sizeof(uint64_t)));
		jb.pool = !c->opt.exact_layout && n_i > 0;
		pr.heavy_skipped = false;
		if (jb.pool) {
			// capacity: last run's exact need + 12 % + room for every wave's partly used slab.  A pair that runs for the first time
			// makes a DRY pass first (same kernels, capacity 0: everything is counted, nothing is written), which the overflow
			// handling below turns into a real pass of the right size.
			pr.dry = pr.need_hint == 0;
			pr.shared_empty = jb.i != jb.j && pr.n_query > 0;
			pr.pooled = true;
			{ const tnsx_status r = size_pool(c, pr, pr.dry ? nullptr : pr.region_payload, pr.region_asked[0] || pr.region_asked[tnsx::POOL_OVERFLOW] ? pr.region_asked : nullptr, false, query_waves); if (r != TNSX_OK) return r; }
			// worklists of the cells the fast / fat kernels pass on (at most one entry per occupied cell)
			const size_t max_cells = (size_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_i, n_cells));
			HIPCHK(c, pr.heavy.reserve(max_cells * sizeof(uint2)));
			HIPCHK(c, pr.heavy2.reserve(max_cells * sizeof(uint2)));
		}
		else {
			HIPCHK(c, pr.counts.reserve((size_t)std::max(n_i, 1) * sizeof(uint32_t)));
			HIPCHK(c, pr.offs_sorted.reserve(((size_t)n_i + 1) * sizeof(uint64_t)));
			HIPCHK(c, c->scan_temp.reserve(tnsx::scan_temp_bytes((size_t)n_i)));
		}
	}

	return TNSX_OK;
}

static tnsx_status enqueue_run_begin(RunAttempt& run)
{
	TNSX_RUN_ALIASES;
	// ---- ONE launch in front of everything (round 4: four kernels of a steady-state step, ~5 us of dispatch each): the words of this attempt and
	//      the occupied-cell counts of the sets that are built start at zero, the table entries the previous run set are cleared, every pool
	//      pass gets the hot words of its control block and its region table
	run.t_build0 = tm.mark();
	{
		tnsx::RunBeginArgs rb{};
		rb.words = d_words; rb.n_words = WB * (size_t)(n_sets + 1); rb.n_occ = c->n_occ.as<uint32_t>();
		for (int si = 0; si < n_sets; si++) {
			PointSet& s = c->sets[si];
			if (keeps_its_build(s)) continue;
			if (si < 64) rb.sets |= 1ull << si;
			else HIPCHK(c, hipMemsetAsync(c->n_occ.as<uint32_t>() + si, 0, sizeof(uint32_t), st));
			// the table is needed even for empty sets (they can be searched into)
			if (sparse) {
				// no table: the block index is written in full by every build; an empty set gets an all-zero index + sentinel here
				const void* old_blk = s.blk.p;
				HIPCHK(c, s.blk.reserve(((size_t)c->sparse_blocks + 2) * sizeof(uint32_t)));
				HIPCHK(c, s.occ.reserve(((size_t)std::max(s.n, 1) + 2) * sizeof(uint2)));
				if (s.n == 0 || s.blk.p != old_blk) HIPCHK(c, hipMemsetAsync(s.blk.p, 0, s.blk.cap, st));
				if (s.n == 0) { const uint2 sent[2] = { { 0u, 0xffffffffu }, { 0u, 0xffffffffu } }; HIPCHK(c, hipMemcpyAsync(s.occ.p, sent, sizeof(sent), hipMemcpyHostToDevice, st)); }
				s.table_state = 2;   // (should the next run be dense again: a table of unknown content)
			}
			else {
				const void* old_table = s.table.p;
				HIPCHK(c, s.table.reserve(n_cells * sizeof(uint2)));
				// (the list-based clear reads the PREVIOUS occupied-cell list inside k_run_begin; launch_build re-reserves that list for this run's build.  A list that
				//  has to grow would be freed while the enqueued kernel may still read it -- correct today only because hipFree synchronises -- so a
				//  set that outgrew its list clears the whole table instead and k_run_begin never sees the old list)
				const bool occ_grows = ((size_t)std::max(s.n, 1) + 2) * sizeof(uint2) > s.occ.cap;
				if (s.table.p != old_table || s.table_state == 2 || (occ_grows && s.table_state == 1 && s.table_dirty > 0))
					HIPCHK(c, hipMemsetAsync(s.table.p, 0, s.table.cap, st));   // new or unknown: all of it
				else if (s.table_state == 1 && s.table_dirty > 0) {
					if (rb.n_clear < tnsx::RUN_BEGIN_MAX_SETS) rb.clear[rb.n_clear++] = { s.occ.as<uint2>(), s.table.as<uint2>(), s.table_dirty };
					else tnsx::launch_table_clear(s.occ.as<uint2>(), s.table_dirty, s.table.as<uint2>(), st);
				}
				s.table_state = 0; s.table_dirty = 0;
			}
			// the bucket build's first pass in one read: the previous build of this set on this grid left a window per bucket (see k_bucket_scatter);
			// an overflowing window raises the guard flag like a point outside the box does
			int nb = 0;
			s.bk_now = !sparse && s.n > 0 && tnsx::cell_build_uses_buckets(s.n, key_bits, c->opt.exact_layout != 0, c->opt.bucket_build_min_points, &nb);
			s.bk_used = false;
			if (s.bk_now) {
				const void* old_win = s.bk_win.p;
				HIPCHK(c, s.bk_win.reserve((size_t)nb * sizeof(uint2)));
				HIPCHK(c, s.bk_cur.reserve((size_t)nb * tnsx::BUCKET_CURSOR_STRIDE * sizeof(uint32_t)));
				if (s.bk_win.p != old_win || s.bk_buckets != nb) s.bk_gen = 0;
				s.bk_used = speculate && c->opt.temporal_reuse != 0 && s.bk_gen == c->grid_gen && s.bk_gen != 0 && rb.n_zero < tnsx::RUN_BEGIN_MAX_SETS &&
				            (int64_t)s.n <= (int64_t)s.bk_n + s.bk_n / 16 && (int64_t)s.n >= (int64_t)s.bk_n - s.bk_n / 16;
				if (s.bk_used) { rb.zero[rb.n_zero] = s.bk_cur.as<uint32_t>(); rb.n_zero_words[rb.n_zero] = (uint32_t)nb * tnsx::BUCKET_CURSOR_STRIDE; rb.n_zero++; }
				s.bk_buckets = nb;
			}
			else s.bk_gen = 0;
		}
		for (size_t k = 0; k < jobs.size() && rb.n_pool < tnsx::RUN_BEGIN_MAX_POOLS; k++) {
			RunJob& jb = jobs[k];
			if (!jb.pool) continue;
			const PairResult& pr = run.pair(jb);
			tnsx::RunBeginPool& bp = rb.pool[rb.n_pool++];
			const bool count_only = c->debug_nostore || pr.dry;
			for (int r = 0; r < PairResult::NR; r++) { bp.regions[2 * r] = pr.region_base[r]; bp.regions[2 * r + 1] = count_only ? 0ull : pr.region_cap[r]; }
			bp.ctrl = ctrl_slot(k, 0); bp.offs = pr.offs_orig.as<uint64_t>(); bp.n_shared_empty = pr.shared_empty ? (size_t)pr.n_query : 0; bp.records = pr.records.as<int>();
			jb.begun = true;
		}
		tnsx::launch_run_begin(rb, st);
	}
	return TNSX_OK;
}

static tnsx_status launch_build(RunAttempt& run)
{
	TNSX_RUN_ALIASES;
	{ const tnsx_status r = enqueue_run_begin(run); if (r != TNSX_OK) return r; }
	for (int si = 0; si < n_sets; si++) {
		PointSet& s = c->sets[si];
		const bool cacheable = !s.user_ids && s.n > 0;
		if (keeps_its_build(s)) {
			// taken to be unchanged: only its checksum is computed (and compared after the run)
			tnsx::launch_set_checksum(s.d_xyz, variable ? s.d_radii : nullptr, s.n, d_words + WB * (size_t)(1 + si), st);
			skipped[(size_t)si] = 1;
			S.n_cached_sets++;
			continue;
		}
		HIPCHK(c, s.occ.reserve(((size_t)std::max(s.n, 1) + 2) * sizeof(uint2)));   // (after the clear, which reads the previous list, has been enqueued; + sentinel)
		s.built_gen = 0;
		if (s.n == 0) continue;
		if (!sparse) s.table_state = 2;   // until this run's occupied-cell count has reached the host
		for (int k = 0; k < 2; k++) {
			// ([1] is the intermediate array of the bucket build: its buckets lie in windows with some slack)
			// (the windows in use were laid out for bk_n points: with fewer points now they still reach as far)
			HIPCHK(c, s.xyzi[k].reserve((k == 1 && s.bk_now ? tnsx::bucket_window_slots(std::max(s.n, s.bk_used ? s.bk_n : 0), s.bk_buckets) : (size_t)s.n) * sizeof(float4)));
			if (variable) HIPCHK(c, s.r2[k].reserve((size_t)s.n * sizeof(float)));
		}
		HIPCHK(c, c->sort_temp.reserve(tnsx::cell_build_temp_bytes(s.n)));
		tnsx::CellSortBuffers cb;
		for (int k = 0; k < 2; k++) { cb.xyzi[k] = s.xyzi[k].as<float4>(); cb.r2[k] = s.r2[k].as<float>(); }
		tnsx::BuildGuard gd;
		if (speculate) {
			for (int d = 0; d < 3; d++) { gd.lo[d] = c->grid_lo[d]; gd.hi[d] = c->grid_hi[d]; }
			gd.r_max = c->grid_r_max;
			gd.flag = reinterpret_cast<uint32_t*>(d_words);
			if (c->grid_trimmed) {
				// the grid covers the bulk of the points only: whoever is outside it is binned into its border cells (exact), so the box that
				// must hold is the WORLD box (whose update needs the true bounds); the points outside the grid's own box are counted, and
				// the grid is laid out afresh when they become many
				for (int d = 0; d < 3; d++) { gd.soft_lo[d] = c->grid_lo[d]; gd.soft_hi[d] = c->grid_hi[d]; gd.lo[d] = c->world[d]; gd.hi[d] = c->world[3 + d]; }
				gd.outside = d_words + tnsx::CHK_STRIDE;
			}
		}
		if (cacheable) gd.checksum = d_words + WB * (size_t)(1 + si);
		if (s.user_ids) HIPCHK(c, s.orig_sorted.reserve((size_t)s.n * sizeof(uint32_t)));
		// sort + cell table + occupied-cell list (two-pass bucket build where the key fits; the exact layout keeps the stable sort)
		int passes = 0;
		const uint32_t q_limit = (s.n_query >= 0 && s.n_query < s.n) ? (uint32_t)s.n_query : 0xffffffffu;
		s.sorted_buf = tnsx::launch_cell_build(s.d_xyz, variable ? s.d_radii : nullptr, s.n, g, key_bits, cb, c->sort_temp.p, s.user_ids,
		                                       s.user_ids ? s.orig_sorted.as<uint32_t>() : nullptr, gd, q_limit, c->opt.exact_layout != 0,
		                                       c->opt.bucket_build_min_points, s.table.as<uint2>(), s.occ.as<uint2>(), c->n_occ.as<uint32_t>() + si, &passes,
		                                       tnsx::BucketWindows{ s.bk_used && gd.flag != nullptr, s.bk_now ? s.bk_win.as<uint2>() : nullptr, s.bk_cur.as<uint32_t>() },
		                                       tnsx::SparseCells{ sparse ? s.blk.as<uint32_t>() : nullptr, c->sparse_shift, c->sparse_blocks }, st);
		S.radix_passes = passes;
		if (s.bk_used) S.one_read_builds++;
	}
	return TNSX_OK;
}

// what the query kernels of pair (jb.i -> jb.j) get
static tnsx::QueryArgs make_query_args(RunAttempt& run, const RunJob& jb, PairResult& pr, size_t k)
{
	TNSX_RUN_ALIASES;
	const PointSet& A = c->sets[jb.i];
	const PointSet& B = c->sets[jb.j];
	tnsx::QueryArgs a{};
	a.occ_i = A.occ.as<uint2>(); a.n_occ_i = c->n_occ.as<uint32_t>() + jb.i;
	a.table_i = A.table.as<uint2>();
	a.xyzi_i = A.xyzi[A.sorted_buf].as<float4>(); a.r2_i = A.r2[A.sorted_buf].as<float>();
	a.orig_i = A.user_ids ? A.orig_sorted.as<uint32_t>() : nullptr;
	a.table_j = B.table.as<uint2>(); a.xyzi_j = B.xyzi[B.sorted_buf].as<float4>(); a.r2_j = B.r2[B.sorted_buf].as<float>();
	a.r2_fixed = c->radius_sq;
	a.query_limit = A.n_query < 0 ? 0xffffffffu : (uint32_t)A.n_query;
	a.n_points_i = (uint32_t)A.n;
	a.g = g;
	a.counts = pr.counts.as<uint32_t>();
	a.offs_sorted = pr.offs_sorted.as<uint64_t>();
	a.records = pr.records.as<int>();
	a.offs_by_orig = pr.offs_orig.as<uint64_t>();
	a.pool_cursor = reinterpret_cast<unsigned long long*>(ctrl_slot(k, tnsx::CTRL_CURSOR));
	a.pool_regions = reinterpret_cast<const unsigned long long*>(ctrl_slot(k, tnsx::CTRL_REGIONS));
	a.tickets = ctrl_slot(k, tnsx::CTRL_TICKETS);
	a.n_heavy = ctrl_slot(k, tnsx::CTRL_NHEAVY);
	a.tickets2 = ctrl_slot(k, tnsx::CTRL_TICKETS2);
	a.n_heavy2 = ctrl_slot(k, tnsx::CTRL_NHEAVY2);
	a.heavy = pr.heavy.as<uint2>();
	a.heavy2 = pr.heavy2.as<uint2>();
	// (the worklist of the group formulation shares buffer and counter with the candidate-presence filter: that one is for pairs of two
	//  different sets, the group formulation for a set searched in itself)
	a.heavy0 = pr.filtered.as<uint2>();
	a.n_heavy0 = ctrl_slot(k, tnsx::CTRL_NFILTERED);
	if (sparse) { a.socc_i = A.occ.as<uint2>(); a.blk_i = A.blk.as<uint32_t>(); a.socc_j = B.occ.as<uint2>(); a.blk_j = B.blk.as<uint32_t>(); a.sparse_shift = c->sparse_shift; }
	a.abort_flag = speculate ? reinterpret_cast<const uint32_t*>(d_words) : nullptr;
	a.pool_slab = pr.pool_slab;
	a.pool_slab_heavy = std::max<uint32_t>(pr.pool_slab, 8192u);
	a.shared_empty = jb.pool && jb.i != jb.j ? 1u : 0u;
	if (a.shared_empty && pr.n_query > 0 && !sparse) {
		// the query walks the cells that have candidates at all (launch_mark_cells + launch_filter_marked, enqueued by launch_pool)
		a.occ_i = pr.filtered.as<uint2>();
		a.n_occ_i = ctrl_slot(k, tnsx::CTRL_NFILTERED);
	}
	else if (pr.dry && pr.sample_stride > 1) {
		// a count-only pass over a SAMPLE of the occupied cells (launch_pool_pass; the list and its length share buffer and counter with the presence filter)
		a.occ_i = pr.filtered.as<uint2>();
		a.n_occ_i = ctrl_slot(k, tnsx::CTRL_NFILTERED);
	}
	return a;
}

// tiers: bit 0 = begin (if the pass was not begun by launch_run_begin) + candidate-presence filter + first tier, bit 1 = the heavy tiers
static tnsx_status launch_pool_pass(RunAttempt& run, size_t k, int tiers, bool fresh)
{
	TNSX_RUN_ALIASES;
	const RunJob& jb = jobs[k];
	PairResult& pr = run.pair(jb);
	if ((tiers & 1) && (fresh || !jb.begun)) {
		// the region table of this pass (all capacities 0: nothing is written, everything is counted).
		// A pair of two different sets: most query cells may have no candidate at all (a fluid searched in its boundary).  Int 0 of
		// the pool is THE empty record and every offset starts out pointing at it: cells without candidates then cost no
		// allocation, no record and no scattered 8-byte offset store.
		unsigned long long regions[2 * PairResult::NR];
		const bool count_only = c->debug_nostore || pr.dry;
		for (int r = 0; r < PairResult::NR; r++) { regions[2 * r] = pr.region_base[r]; regions[2 * r + 1] = count_only ? 0ull : pr.region_cap[r]; }
		tnsx::launch_pool_begin(regions, ctrl_slot(k, 0), pr.offs_orig.as<uint64_t>(),
		                        pr.shared_empty ? (size_t)pr.n_query : 0, pr.records.as<int>(), st);
	}
	const int t0 = tm.mark();   // (behind the last kernel of the build / of the previous pass: the bracket holds the pass's query kernels only)
	if (pr.shared_empty && (tiers & 1) && !sparse) {   // (the presence filter's byte map is a dense structure: a sparse grid walks all occupied cells)
		const PointSet& A = c->sets[jb.i];
		const size_t max_cells = (size_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)pr.n_i, n_cells));
		HIPCHK(c, pr.filtered.reserve(max_cells * sizeof(uint2)));
		{
			const PointSet& B = c->sets[jb.j];
			const void* old_map = c->cell_map.p;
			HIPCHK(c, c->cell_map.reserve(n_cells));
			if (c->cell_map.p != old_map) HIPCHK(c, hipMemsetAsync(c->cell_map.p, 0, c->cell_map.cap, st));   // (all zero between uses)
			const size_t max_cells_j = (size_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(B.n, 1), n_cells));
			unsigned char* map = c->cell_map.as<unsigned char>();
			tnsx::launch_mark_cells(B.occ.as<uint2>(), c->n_occ.as<uint32_t>() + jb.j, g, map, 1, max_cells_j, st);
			tnsx::launch_filter_marked(A.occ.as<uint2>(), c->n_occ.as<uint32_t>() + jb.i, map, pr.filtered.as<uint2>(), ctrl_slot(k, tnsx::CTRL_NFILTERED), max_cells, st);
			tnsx::launch_mark_cells(B.occ.as<uint2>(), c->n_occ.as<uint32_t>() + jb.j, g, map, 0, max_cells_j, st);
		}
	}
	// Round 6: the count-only pass that sizes the pool of a pair's first run looks at every 32nd occupied cell of a large set (C2: 25 k of 804 k cells, 1/32 of the
	// 1.45 ms a full count costs).  The list is in key order and an XCD works on a fixed eighth of it, so the sample of an eighth is the sample of that XCD's region;
	// the counts are scaled by the stride (collect_pair) and the sized pass keeps its margins (a quarter / a half more than counted) and its repair.
	pr.sample_stride = 1;
	if (pr.dry && (tiers & 1) && !pr.shared_empty && !sparse && pr.n_i >= (1 << 20) && !(c->opt.query_formulation == 1 && tnsx_query_formulation_available(1) != 0)) {
		const PointSet& A = c->sets[jb.i];
		pr.sample_stride = 32;
		const size_t max_cells = (size_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)pr.n_i, n_cells));
		HIPCHK(c, pr.filtered.reserve((max_cells / pr.sample_stride + 2) * sizeof(uint2)));
		tnsx::launch_sample_cells(A.occ.as<uint2>(), c->n_occ.as<uint32_t>() + jb.i, pr.sample_stride, pr.filtered.as<uint2>(), ctrl_slot(k, tnsx::CTRL_NFILTERED),
		                          max_cells / pr.sample_stride + 1, st);
	}
	if (pr.n_i > 0) {
		qc.self = jb.i == jb.j; qc.mode = tnsx::QUERY_POOL;
		// opt-in (tnsx_options.query_formulation = 1), fixed radius, a set searched in itself: the group formulation (tools/ubench/tnsx_query_group.hip, variant builds only)
		// in front of the cell kernels, unless it was switched off for this pair
		pr.groups_now = !sparse && !variable && jb.i == jb.j && c->opt.query_formulation == 1 && tnsx_query_formulation_available(1) != 0 && !pr.groups_off &&
		                c->grid_h * c->grid_h > 1e-30f;
		qc.groups = pr.groups_now;
		if (pr.groups_now) { HIPCHK(c, pr.filtered.reserve((size_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)pr.n_i, n_cells)) * sizeof(uint2))); tiers = 3; }
		qc.tiers = tiers;
		tnsx::launch_query(make_query_args(run, jb, pr, k), qc, c->n_cus, st);
		qc.groups = false; qc.tiers = 3;
	}
	pr.heavy_skipped = !(tiers & 2);
	const int t1 = tm.mark();
	span(ST_FILL, t0, t1);
	uint32_t* const h_count = (pr.shared_empty && !sparse) ? h_filt + k : (pr.groups_now && pr.n_i > 0 ? h_left + k : nullptr);   // (the two worklists share a counter, see make_args)
	if (defer_readback && run_end.n_jobs < tnsx::RUN_END_MAX_JOBS) {
		tnsx::RunEndJob& rj = run_end.job[run_end.n_jobs++];
		rj.ctrl_cursor = ctrl_slot(k, tnsx::CTRL_CURSOR); rj.h_ctrl = reinterpret_cast<unsigned long long*>(h_ctrl + HC * k);
		rj.d_count = ctrl_slot(k, tnsx::CTRL_NFILTERED); rj.h_count = h_count;
		rj.d_heavy = ctrl_slot(k, tnsx::CTRL_NHEAVY); rj.h_heavy = h_heavy + k;
		return TNSX_OK;
	}
	// (one plain copy per region: the first hipMemcpy2DAsync of a process costs 6.7 ms -- it loads the runtime's blit kernels -- and this is the cold run's path)
	for (int r = 0; r < PairResult::NR; r++)
		HIPCHK(c, hipMemcpyAsync(h_ctrl + HC * k + (size_t)r * tnsx::POOL_CTRL_WORDS, ctrl_slot(k, tnsx::CTRL_CURSOR) + (size_t)r * tnsx::CTRL_STRIDE_U32,
		                         tnsx::POOL_CTRL_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
	if (h_count) HIPCHK(c, hipMemcpyAsync(h_count, ctrl_slot(k, tnsx::CTRL_NFILTERED), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	HIPCHK(c, hipMemcpyAsync(h_heavy + k, ctrl_slot(k, tnsx::CTRL_NHEAVY), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	return TNSX_OK;
}

static tnsx_status launch_queries(RunAttempt& run)
{
	TNSX_RUN_ALIASES;
	qc = tnsx::QueryConfig{};
	qc.arith = c->opt.arith;
	qc.variable = variable;
	qc.symmetric = variable && c->symmetric;   // TreeNSearch.cpp:2431
	qc.blocks_per_cu = c->opt.query_blocks_per_cu;
	qc.fast_blocks_per_cu = c->opt.fast_blocks_per_cu;
	// (stage times: the build is everything from the first launch of the attempt to here -- table clear and control blocks included; no event
	//  between its kernels, every event record between two kernels is a bubble of 6-9 us)
	const int t_build1 = tm.mark();
	span(ST_SORT, run.t_build0, t_build1);
	for (size_t k = 0; k < jobs.size(); k++) {
		RunJob& jb = jobs[k];
		PairResult& pr = run.pair(jb);
		const int n_i = pr.n_i;
		if (jb.pool) {
			// the heavy tiers (cells with more than 512 candidates or more than 64 query points) are not launched when the previous run of the
			// pair had nothing for them: what the first tier passes on is counted, and they run after the synchronisation if it did
			const bool light = speculate && !pr.dry && pr.heavy_known && pr.heavy_cells == 0;
			const tnsx_status r = launch_pool_pass(run, k, light ? 1 : 3, false);
			if (r != TNSX_OK) return r;
		}
		else {
			const int t0 = tm.mark();
			if (n_i > 0) {
				qc.self = jb.i == jb.j; qc.mode = tnsx::QUERY_COUNT;
				const tnsx::QueryArgs qa = make_query_args(run, jb, pr, k);
				tnsx::launch_exact_nan(qa.xyzi_i, qa.orig_i, n_i, qa.query_limit, qa.counts, nullptr, nullptr, nullptr, 0, st);   // (points that enter no cell: NaN x)
				tnsx::launch_query(qa, qc, c->n_cus, st);
			}
			const int t1 = tm.mark();
			tnsx::exclusive_scan_u32_to_u64(pr.counts.as<uint32_t>(), pr.offs_sorted.as<uint64_t>(), (size_t)n_i, c->scan_temp.p, st);
			HIPCHK(c, hipMemcpyAsync(h_ctrl + HC * k, pr.offs_sorted.as<uint64_t>() + n_i, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
			const int t2 = tm.mark();
			span(ST_COUNT, t0, t1); span(ST_SCAN, t1, t2);
		}
	}
	run_end.n_occ = c->n_occ.as<uint32_t>(); run_end.h_nocc = h_nocc; run_end.n_sets = n_sets;
	run_end.words = d_words; run_end.h_words = reinterpret_cast<unsigned long long*>(h_words); run_end.n_words = WB * (size_t)(n_sets + 1);
	tnsx::launch_run_end(run_end, st);
	defer_readback = false;
	return TNSX_OK;
}

static tnsx_status judge_attempt(RunAttempt& run, bool* redo)
{
	TNSX_RUN_ALIASES;
	{ const tnsx_status r = sync_stream(c); if (r != TNSX_OK) return r; }   // record totals / pool cursors / what was speculated on are needed on the host

	// ---- were the assumptions of this attempt right?
	for (int si = 0; si < n_sets; si++) {
		PointSet& s = c->sets[si];
		if (s.n > 0 && !sparse) { s.table_state = 1; if (!skipped[(size_t)si]) s.table_dirty = h_nocc[si]; }
	}
	bool wrong = speculate && (h_words[0] & 0xffffffffull) != 0;   // a point left the box of the grid / a radius outgrew its cell edge
	if (wrong) c->grid_valid = false;
	// a reused trimmed grid: results are exact whatever lies outside it, but more than 0.2 % of the points in its border cells is the
	// sign that the bulk has moved -- the next run lays the grid out afresh
	if (speculate && c->grid_trimmed && h_words[tnsx::CHK_STRIDE] > (uint64_t)std::max<int64_t>(16, n_total >> 9)) c->grid_valid = false;
	for (int si = 0; si < n_sets; si++) {
		PointSet& s = c->sets[si];
		const bool cacheable = !s.user_ids && s.n > 0;
		const bool same_input = s.chk_valid && s.chk_xyz == s.user_xyz && s.chk_radii == s.user_radii && s.chk_n == s.n && s.chk_double == s.is_double;
		uint64_t chk_now = 0;
		for (int k = 0; k < tnsx::CHK_SLOTS; k++) chk_now += h_words[WB * (size_t)(1 + si) + (size_t)k * tnsx::CHK_STRIDE];
		const bool unchanged = cacheable && same_input && s.chk_value == chk_now;
		if (skipped[(size_t)si] && !unchanged) { wrong = true; s.predicted_static = false; s.chk_valid = false; continue; }
		if (wrong) continue;                      // (an attempt that is thrown away teaches nothing)
		s.predicted_static = unchanged;           // two equal checksums in a row: the next run keeps the structures
		s.chk_valid = cacheable; s.chk_value = chk_now;
		s.chk_xyz = s.user_xyz; s.chk_radii = s.user_radii; s.chk_n = s.n; s.chk_double = s.is_double;
		if (!skipped[(size_t)si] && s.n > 0) s.built_gen = c->grid_gen;
		if (!skipped[(size_t)si] && s.bk_now) { s.bk_gen = c->grid_gen; s.bk_n = s.n; }   // (this build wrote the windows of the next one)
	}
	if (wrong) { *redo = true; S.speculation_redos++; return TNSX_OK; }
	return TNSX_OK;
}

// the outcome of the pass of pair k: its counters, the heavy tiers if they were left out and turn out to be needed, the repeat of a pass whose pool overflowed (or
// that was a dry pass), the exact layout's fill pass
static tnsx_status collect_pair(RunAttempt& run, size_t k)
{
	TNSX_RUN_ALIASES;
	const RunJob& jb = jobs[k];
	PairResult& pr = run.pair(jb);
	uint64_t n_neighbors = 0;
	if (jb.pool) {
		// what every XCD produced: ints it asked for - ints it left unused.  The pass failed if a wave found both its own region and the
		// overflow region full (or if it was a dry pass): size the pool by what was counted and redo this pair's pass.
		const uint64_t* hc = h_ctrl + HC * k;
		if (pr.heavy_skipped && h_heavy[k] != 0u) {
			// the first tier did pass cells on after all: the two heavy tiers now, on the same control block
			const tnsx_status r = launch_pool_pass(run, k, 2, false);
			if (r != TNSX_OK) return r;
			HIPCHK(c, hipStreamSynchronize(st));
			S.heavy_catchups++;
		}
		pr.heavy_cells = h_heavy[k]; pr.heavy_known = true;
		uint64_t payload[PairResult::NR], asked_now[PairResult::NR];
		auto read_counters = [&]() {
			n_neighbors = 0;
			for (int r = 0; r < PairResult::NR; r++) {
				const uint64_t asked = hc[(size_t)r * tnsx::POOL_CTRL_WORDS];
				asked_now[r] = asked;
				// (the common region: what the heavy tiers asked for + what fast-tier waves were diverted to it, the latter counted in their
				//  own region as well -- a harmless overestimate in the rare run where a region was full)
				payload[r] = asked - std::min(asked, hc[(size_t)r * tnsx::POOL_CTRL_WORDS + tnsx::POOL_WASTE_WORD]);
				n_neighbors += hc[(size_t)r * tnsx::POOL_CTRL_WORDS + tnsx::POOL_HITS_WORD];
			}
		};
		read_counters();
		for (int attempt = 0; !c->debug_nostore && (pr.dry || hc[(size_t)tnsx::POOL_OVERFLOW * tnsx::POOL_CTRL_WORDS] > pr.region_cap[tnsx::POOL_OVERFLOW]); attempt++) {
			if (attempt >= 5) TNSX_FAIL(c, TNSX_ERR_HIP, "neighbour pool kept overflowing (%llu neighbours)", (unsigned long long)n_neighbors);
			const bool was_dry = pr.dry;
			if (pr.dry) {
				pr.dry = false;
				if (pr.sample_stride > 1) {
					// the pass looked at every sample_stride-th cell: scale what it counted (the margins of a pool sized after a dry pass cover the sampling error)
					for (int r = 0; r < PairResult::NR; r++) payload[r] *= pr.sample_stride;
					S.sampled_passes++;
					pr.sample_stride = 1;
				}
				else S.cold_passes++;   // the dry pass counted everything: the real pass is sized exactly
			}
			else {
#ifdef TNSX_BUILD_DEBUG_POOL
				fprintf(stderr, "[tnsx] pool overflow pair %zu: overflow region asked %llu of %llu, slab %u n_i %d\n", k,
				                                            (unsigned long long)hc[(size_t)tnsx::POOL_OVERFLOW * tnsx::POOL_CTRL_WORDS], (unsigned long long)pr.region_cap[tnsx::POOL_OVERFLOW], pr.pool_slab, pr.n_i);
#endif
				S.pool_retries++;
			}
			{ const tnsx_status r = size_pool(c, pr, payload, was_dry ? nullptr : asked_now, !was_dry, query_waves); if (r != TNSX_OK) return r; }
			const tnsx_status r = launch_pool_pass(run, k, 3, true);
			if (r != TNSX_OK) return r;
			HIPCHK(c, hipStreamSynchronize(st));
			pr.heavy_cells = h_heavy[k];
			read_counters();
		}
		pr.n_records = 1;
		uint64_t sum = 0;
		for (int r = 0; r < PairResult::NR; r++) {
			pr.region_payload[r] = payload[r];
			pr.region_asked[r] = asked_now[r];
			sum += payload[r];
			pr.region_used[r] = c->debug_nostore ? 0 : std::min<uint64_t>(hc[(size_t)r * tnsx::POOL_CTRL_WORDS], pr.region_cap[r]);
			if (pr.region_used[r]) pr.n_records = std::max(pr.n_records, pr.region_base[r] + pr.region_used[r]);
		}
		pr.need_hint = sum + 1;
		// A query point that entered no cell (a NaN x is "no point") was never visited, so nobody wrote its offset.  The records that were written say
		// whether there is one: ints of records = neighbours + one count word per visited query.  Rare; its offsets then go to the empty record, so that
		// every consumer -- the device views, the compaction of the host mirror -- finds a list of length 0 there (round-5 advice).
		if (!pr.shared_empty && !c->debug_nostore && pr.n_query > 0 && sum != n_neighbors + (uint64_t)pr.n_query) {
			const PointSet& A = c->sets[jb.i];
			tnsx::launch_point_nan_offsets(A.d_xyz, pr.n_query, pr.offs_orig.as<uint64_t>(), st);
			S.nan_fixups++;
		}
	}
	else {
		pr.n_records = h_ctrl[HC * k];
		n_neighbors = pr.n_records - (uint64_t)pr.n_query;
		pr.shared_empty = false;
		pr.pooled = false;
		for (int r = 0; r < PairResult::NR; r++) { pr.region_base[r] = 0; pr.region_used[r] = 0; }
		pr.region_used[0] = pr.n_records;
		HIPCHK(c, pr.records.reserve(std::max<uint64_t>(pr.n_records, 1) * sizeof(int)));
		const int t0 = tm.mark();
		if (pr.n_i > 0) {
			qc.self = jb.i == jb.j; qc.mode = tnsx::QUERY_FILL;
			const tnsx::QueryArgs qa = make_query_args(run, jb, pr, k);
			tnsx::launch_query(qa, qc, c->n_cus, st);
			tnsx::launch_exact_nan(qa.xyzi_i, qa.orig_i, pr.n_i, qa.query_limit, nullptr, qa.offs_sorted, qa.records, qa.offs_by_orig, 1, st);
		}
		const int t1 = tm.mark();
		span(ST_FILL, t0, t1);
	}
	pr.n_neighbors = n_neighbors;
	pr.valid = true;
	S.n_queries += (uint64_t)pr.n_query;
	S.n_neighbors += n_neighbors;
	if (jb.pool) S.n_pool_pairs++;
	S.n_filtered_cells += h_filt[k];
	pr.n_cells_i = h_nocc[jb.i];
	if (jb.pool && pr.groups_now) {
		S.n_group_pairs++;
		S.n_group_passed_cells += h_left[k];
		// more than a quarter of the occupied cells passed on (dense cells, lists longer than the lanes' capacity): the cell kernels
		// alone are the better tool for this pair
		if ((uint64_t)h_left[k] * 4u > (uint64_t)h_nocc[jb.i] + 64u) pr.groups_off = true;
	}
	return TNSX_OK;
}

static tnsx_status finish_run(RunAttempt& run)
{
	TNSX_RUN_ALIASES;
	for (size_t k = 0; k < jobs.size(); k++) { const tnsx_status r = collect_pair(run, k); if (r != TNSX_OK) return r; }
	for (int si = 0; si < n_sets; si++) S.n_occupied_cells += h_nocc[si];

	// ---- optional: ascending order inside every record
	if (c->opt.sorted_lists) {
		const int t0 = tm.mark();
		for (const RunJob& jb : jobs) {
			PairResult& pr = run.pair(jb);
			tnsx::launch_sort_records(pr.records.as<int>(), pr.offs_orig.as<uint64_t>(), pr.n_query, c->n_cus, st);
		}
		const int t1 = tm.mark();
		span(ST_SORT_LISTS, t0, t1);
	}

	// ---- optional pinned host mirror (what get_neighborlist needs on the CPU side)
	const int e_m0 = tm.mark();
	if (c->opt.mirror_to_host) {
		for (const RunJob& jb : jobs) {
			PairResult& pr = run.pair(jb);
			{ const tnsx_status r = mirror_pair(c, pr, st); if (r != TNSX_OK) return r; }
		}
	}
	const int e_end = tm.mark();
	span(ST_MIRROR, e_m0, e_end);
	{ const tnsx_status r = sync_stream(c); if (r != TNSX_OK) return r; }   // run() is synchronous like the reference

	// ---- statistics: algorithmic bytes with the measured Q, E, C
	{
		const uint64_t N = S.n_points, Q = S.n_queries, E = S.n_neighbors, C = S.n_occupied_cells;
		const uint64_t rho = variable ? 4 : 0;
		const uint64_t P = (uint64_t)S.radix_passes;
		S.bytes_build = (68 + rho + 16 * P) * N + 8 * C;
		S.bytes_query = 16 * N + 16 * Q + 4 * E;
	}
	if (c->opt.collect_stage_times) {
		float acc[ST_N] = { 0 };
		for (const RunSpan& sp : run.spans) acc[sp.stage] += tm.ms(sp.a, sp.b);
		S.ms_upload = acc[ST_UPLOAD]; S.ms_bounds = acc[ST_BOUNDS]; S.ms_table_clear = acc[ST_KEYS]; S.ms_sort = acc[ST_SORT];
		S.ms_cells = acc[ST_CELLS]; S.ms_count = acc[ST_COUNT]; S.ms_scan = acc[ST_SCAN];
		S.ms_fill = acc[ST_FILL]; S.ms_mirror = acc[ST_MIRROR]; S.ms_sort_lists = acc[ST_SORT_LISTS];
		S.ms_total = tm.ms(run.e_begin, e_end);
	}
	c->ran = true;
	c->cells_valid = true;
	return TNSX_OK;
}

// one attempt: speculate = the previous run's grid and the sets that did not change are taken over unseen and verified on the device while the attempt
// proceeds; *redo = an assumption was wrong (the caller repeats the run without speculation)
static tnsx_status run_once(tnsx_context* c, bool speculate, bool* redo)
{
	*redo = false;
	RunAttempt run(c, speculate);
	{ const tnsx_status r = plan_run(run); if (r != TNSX_OK) return r; }
	{ const tnsx_status r = launch_build(run); if (r != TNSX_OK) return r; }
	{ const tnsx_status r = launch_queries(run); if (r != TNSX_OK) return r; }
	{ const tnsx_status r = judge_attempt(run, redo); if (r != TNSX_OK || *redo) return r; }
	return finish_run(run);
}


tnsx_status tnsx_run_scalar(tnsx_context* c)
{
	if (!c) return TNSX_ERR_INVALID;
	if (c->multi) return tnsx_run(c);   // (multi-device contexts keep one world box, the one of run())
	c->scalar_world_box = true;
	const tnsx_status r = tnsx_run(c);
	c->scalar_world_box = false;
	return r;
}

tnsx_status tnsx_run(tnsx_context* c)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_MULTI(tnsx_multi::run(c->multi, c->last_error));
	if (hipSetDevice(c->device) != hipSuccess) TNSX_FAIL(c, TNSX_ERR_HIP, "hipSetDevice failed");
	c->stats.speculation_redos = 0;
	// the previous run's grid can be laid over this run's points unseen if nothing it was derived from has changed on the host side
	bool speculate = c->opt.temporal_reuse != 0 && c->grid_valid && c->cell_size > 0.0f && c->grid_variable == !c->radius_set &&
	                 (c->radius_set ? c->grid_r_max == c->radius : true) &&
	                 // run() after run_scalar() (or the other way round) on unchanged points: the reference's two paths snap the world box differently
	                 // (TreeNSearch.cpp:415-472 vs :523-592) -- the box must be looked at again, so no speculation across a change of flavour
	                 c->grid_box_scalar == c->scalar_world_box;
	int64_t n_total = 0;
	for (const PointSet& s : c->sets) n_total += s.n;
	if (n_total == 0) speculate = false;
	for (int attempt = 0; attempt < 3; attempt++) {
		bool redo = false;
		const tnsx_status r = run_once(c, speculate, &redo);
		if (r != TNSX_OK) { c->grid_valid = false; return r; }
		if (!redo) return TNSX_OK;
		speculate = false;
	}
	TNSX_FAIL(c, TNSX_ERR_STATE, "run(): the speculative build kept failing its validation");
}

// ------------------------------------------------------------------------------------------------ results
static tnsx_status find_pair(tnsx_context* c, int i, int j, PairResult** out)
{
	if (!c->ran) TNSX_FAIL(c, TNSX_ERR_STATE, "neighbour lists requested before a successful run()");
	const int n = c->n_sets_at_last_run;
	if (i < 0 || j < 0 || i >= n || j >= n) TNSX_FAIL(c, TNSX_ERR_INVALID, "TreeNSearch::get_neighborlist error: Set does not exist.");
	PairResult& pr = c->pairs[(size_t)i * n + j];
	if (!pr.valid) TNSX_FAIL(c, TNSX_ERR_STATE, "TreeNSearch::get_neighborlist error: Set pair not active.");
	*out = &pr;
	return TNSX_OK;
}

tnsx_status tnsx_mirror_pair_to_host(tnsx_context* c, int i, int j)
{
	if (!c) return TNSX_ERR_INVALID;
	if (c->multi) { tnsx_csr_view v; return tnsx_multi::pair_view(c->multi, i, j, &v, c->last_error); }   // (always mirrored)
	PairResult* pr = nullptr;
	{ const tnsx_status r = find_pair(c, i, j, &pr); if (r != TNSX_OK) return r; }
	std::lock_guard<std::mutex> lock(c->mirror_mutex);
	if (pr->mirrored) return TNSX_OK;
	if (hipSetDevice(c->device) != hipSuccess) TNSX_FAIL(c, TNSX_ERR_HIP, "hipSetDevice failed");
	{ const tnsx_status r = mirror_pair(c, *pr, c->stream); if (r != TNSX_OK) return r; }
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return TNSX_OK;
}

tnsx_status tnsx_get_pair_view(tnsx_context* c, int i, int j, tnsx_csr_view* out)
{
	if (!c || !out) return TNSX_ERR_INVALID;
	TNSX_MULTI(tnsx_multi::pair_view(c->multi, i, j, out, c->last_error));
	PairResult* pr = nullptr;
	{ const tnsx_status r = find_pair(c, i, j, &pr); if (r != TNSX_OK) return r; }
	out->n_points = pr->n_query;
	out->n_records = pr->n_records;
	out->n_neighbors = pr->n_neighbors;
	out->offsets_device = pr->offs_orig.as<uint64_t>();
	out->records_device = pr->records.as<int>();
	out->offsets_host = pr->mirrored ? pr->h_offs.as<uint64_t>() : nullptr;
	out->records_host = pr->mirrored ? pr->h_records.as<int>() : nullptr;
	return TNSX_OK;
}

tnsx_status tnsx_copy_pair(tnsx_context* c, int i, int j, uint64_t* offsets_dst, int* records_dst, int dst_on_device)
{
	if (!c) return TNSX_ERR_INVALID;
	if (c->multi) {
		if (dst_on_device) TNSX_FAIL(c, TNSX_ERR_STATE, "tnsx_copy_pair to device memory: not available on a multi-device context");
		tnsx_csr_view v;
		const tnsx_status r = tnsx_multi::pair_view(c->multi, i, j, &v, c->last_error);
		if (r != TNSX_OK) return r;
		if (offsets_dst && v.n_points > 0) std::memcpy(offsets_dst, v.offsets_host, (size_t)v.n_points * sizeof(uint64_t));
		if (records_dst && v.n_records > 0) std::memcpy(records_dst, v.records_host, v.n_records * sizeof(int));
		return TNSX_OK;
	}
	PairResult* pr = nullptr;
	{ const tnsx_status r = find_pair(c, i, j, &pr); if (r != TNSX_OK) return r; }
	if (hipSetDevice(c->device) != hipSuccess) TNSX_FAIL(c, TNSX_ERR_HIP, "hipSetDevice failed");
	const hipMemcpyKind kind = dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
	if (offsets_dst && pr->n_query > 0) HIPCHK(c, hipMemcpyAsync(offsets_dst, pr->offs_orig.p, (size_t)pr->n_query * sizeof(uint64_t), kind, c->stream));
	if (records_dst && pr->n_records > 0) { const tnsx_status r = copy_records(c, *pr, records_dst, kind, c->stream); if (r != TNSX_OK) return r; }
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return TNSX_OK;
}

// SURVEY.md section 8(f)4 "CSR tensors out": the pair as a standard gap-free CSR in point order, built ON THE DEVICE into caller memory (device pointers):
// offsets_out[p] .. offsets_out[p + 1] delimit the neighbours of point p in indices_out (n_points + 1 offsets, n_neighbors indices, no count words).
// What get_neighborlist hands out one point at a time (TreeNSearch.cpp:241-249), for consumers that live on the GPU.
tnsx_status tnsx_pair_csr_device(tnsx_context* c, int i, int j, int64_t* offsets_out, int* indices_out)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_NOT_MULTI("tnsx_pair_csr_device");
	PairResult* pr = nullptr;
	{ const tnsx_status r = find_pair(c, i, j, &pr); if (r != TNSX_OK) return r; }
	if (!offsets_out) TNSX_FAIL(c, TNSX_ERR_INVALID, "tnsx_pair_csr_device: null offsets");
	if (hipSetDevice(c->device) != hipSuccess) TNSX_FAIL(c, TNSX_ERR_HIP, "hipSetDevice failed");
	const size_t nq = (size_t)std::max(pr->n_query, 0);
	std::lock_guard<std::mutex> lock(c->mirror_mutex);   // (m_len is the mirror's scratch)
	if (nq == 0) { HIPCHK(c, hipMemsetAsync(offsets_out, 0, sizeof(int64_t), c->stream)); }
	else {
		HIPCHK(c, c->m_len.reserve(nq * sizeof(uint32_t)));
		HIPCHK(c, c->scan_temp.reserve(tnsx::scan_temp_bytes(nq)));
		tnsx::launch_record_lengths(pr->records.as<int>(), pr->offs_orig.as<uint64_t>(), (int)nq, c->m_len.as<uint32_t>(), true, c->stream);
		tnsx::exclusive_scan_u32_to_u64(c->m_len.as<uint32_t>(), reinterpret_cast<uint64_t*>(offsets_out), nq, c->scan_temp.p, c->stream);
		if (indices_out && pr->n_neighbors > 0)
			tnsx::launch_compact_records(pr->records.as<int>(), pr->offs_orig.as<uint64_t>(), reinterpret_cast<const uint64_t*>(offsets_out), (int)nq, indices_out, true, c->stream);
	}
	HIPCHK(c, hipGetLastError());
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return TNSX_OK;
}

tnsx_status tnsx_translate_neighbors(tnsx_context* c, int i, int j, const int* id_map_dev)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_NOT_MULTI("tnsx_translate_neighbors");
	PairResult* pr = nullptr;
	{ const tnsx_status r = find_pair(c, i, j, &pr); if (r != TNSX_OK) return r; }
	if (!id_map_dev) TNSX_FAIL(c, TNSX_ERR_INVALID, "tnsx_translate_neighbors: null id map");
	if (hipSetDevice(c->device) != hipSuccess) TNSX_FAIL(c, TNSX_ERR_HIP, "hipSetDevice failed");
	tnsx::launch_translate_records(pr->records.as<int>(), pr->offs_orig.as<uint64_t>(), pr->n_query, id_map_dev, c->n_cus, c->stream);
	HIPCHK(c, hipGetLastError());
	HIPCHK(c, hipStreamSynchronize(c->stream));
	pr->mirrored = false;   // a host mirror made before holds the untranslated indices
	return TNSX_OK;
}

// ------------------------------------------------------------------------------------------------ zsort
tnsx_status tnsx_prepare_zsort(tnsx_context* c)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_MULTI(tnsx_multi::prepare_zsort(c->multi, c->last_error));
	if (hipSetDevice(c->device) != hipSuccess) TNSX_FAIL(c, TNSX_ERR_HIP, "hipSetDevice failed");
	hipStream_t st = c->stream;
	// _set_up (TreeNSearch.cpp:2584) + world box (TreeNSearch.cpp:2666)
	{ const tnsx_status r = stage_inputs(c); if (r != TNSX_OK) return r; }
	float b8[8] = { FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX, FLT_MAX, -FLT_MAX };
	// (after a run() neither the cell size nor the world box is touched here -- the reference orders the cells of its last run, TreeNSearch.cpp:2603 --
	//  so the bounds pass and its host round trip are skipped: 0.3 ms of a 50 M-point step that calls this every step)
	if (!(c->cells_valid && c->cell_size > 0.0f)) { const tnsx_status r = compute_bounds(c, b8); if (r != TNSX_OK) return r; }
	if (c->cell_size < 0.0f) {
		// _set_up's default (TreeNSearch.cpp:300-316); prepare_zsort does not run _check
		if (!c->radius_set && c->n_sets_with_radii != (int)c->sets.size()) {
			TNSX_FAIL(c, TNSX_ERR_CONFIG, "TreeNSearch error: not all point sets have per-point search radius specified.");
		}
		const tnsx_status r = setup_and_check(c, b8);
		if (r != TNSX_OK) return r;
	}
	int64_t n_total = 0;
	for (const PointSet& s : c->sets) n_total += s.n;
	// (the reference updates the box on its no-tree path only, TreeNSearch.cpp:2671-2674, always through the SIMD version)
	if (n_total > 0 && !c->cells_valid) { const tnsx_status r = update_world_box(c, b8, true); if (r != TNSX_OK) return r; }
	int n_pow2 = std::max(c->world_cells_pow2, 1);
	float zs_inv_h = c->cell_size_inv;
	if (!c->cells_valid && n_total > 0) {
		// No run() since the sets last changed: the reference cannot reuse its tree and sorts the POINTS on the cell grid refined by
		// the largest power of two that keeps it below 2^21 cells per axis (_compute_zsort_order_notree, TreeNSearch.cpp:2678-2699).
		// After a run() it orders whole CELLS and keeps the order inside a cell -- the branch above, and what a stable sort on the
		// cell's Morton code gives.
		const float world_size = c->world[3] - c->world[0];
		float cs = c->cell_size;
		int refine = 1;
		while (world_size / (cs / 2.0f) < 2097151.0f && refine < (1 << 20)) { cs /= 2.0f; refine *= 2; }
		zs_inv_h = 1.0f / cs;
		n_pow2 = (int)std::min<long long>((long long)n_pow2 * refine, 2097152ll);
	}
	c->zsort_inv_h = zs_inv_h;
	const int bits_per_axis = std::max(1, ceil_log2_u64((uint64_t)n_pow2));
	const int key_bits = 3 * bits_per_axis;

	tnsx::GridParams mg{};
	mg.ox = c->world[0]; mg.oy = c->world[1]; mg.oz = c->world[2];
	mg.inv_h = zs_inv_h;
	mg.nx = mg.ny = mg.nz = n_pow2;
	for (PointSet& s : c->sets) {
		s.zsort_n = s.n;
		s.built_gen = 0;               // (the ping-pong arrays of the search structure are the scratch of this sort)
		s.zsort_host.clear();          // fetched from the device when somebody asks for it (get_zsort_order, host-side apply_zsort)
		s.zsort_ready = true;
		if (s.n == 0) continue;
		for (int k = 0; k < 2; k++) HIPCHK(c, s.xyzi[k].reserve((size_t)s.n * sizeof(float4)));
		HIPCHK(c, c->sort_temp.reserve(std::max(tnsx::cell_sort_temp_bytes(s.n), tnsx::zsort_temp_bytes(s.n, key_bits))));
		HIPCHK(c, s.zsort_dev.reserve((size_t)s.n * sizeof(int)));
		// Morton key of the point's cell on the reference grid (cell-level order, stable => deterministic): the same
		// point-moving radix sort as the search structure, the order is the index column of the sorted points
		tnsx::CellSortBuffers cb;
		for (int k = 0; k < 2; k++) { cb.xyzi[k] = s.xyzi[k].as<float4>(); cb.r2[k] = nullptr; }
		(void)tnsx::launch_morton_sort(s.d_xyz, s.n, mg, key_bits, cb, c->sort_temp.p, s.zsort_dev.as<int>(), st);
		HIPCHK(c, hipGetLastError());
	}
	HIPCHK(c, hipStreamSynchronize(st));
	c->cells_valid = false;   // TreeNSearch.cpp:2659-2660: the cells are no longer in index order
	// xyzi[] of the search structures was reused as scratch: results of the previous run() stay valid (they do not
	// depend on it), and the next run() rebuilds everything anyway.
	return TNSX_OK;
}

// host copy of the z-order of one set, made on first use
static tnsx_status zsort_host_order(tnsx_context* c, PointSet& s)
{
	if (s.zsort_host.size() == (size_t)s.zsort_n) return TNSX_OK;
	s.zsort_host.assign((size_t)s.zsort_n, 0);
	if (s.zsort_n > 0) {
		if (hipSetDevice(c->device) != hipSuccess) TNSX_FAIL(c, TNSX_ERR_HIP, "hipSetDevice failed");
		HIPCHK(c, hipMemcpyAsync(s.zsort_host.data(), s.zsort_dev.p, (size_t)s.zsort_n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
	}
	return TNSX_OK;
}

tnsx_status tnsx_get_zsort_order(tnsx_context* c, int set_i, const int** host, const int** dev, int* n)
{
	if (!c) return TNSX_ERR_INVALID;
	if (c->multi) { if (dev) *dev = nullptr; return tnsx_multi::zsort_order(c->multi, set_i, host, n, c->last_error); }
	if (!set_ok(c, set_i)) TNSX_FAIL(c, TNSX_ERR_INVALID, "tns::TreeNSearch::apply_zsort error: set to z_sort does not exit.");
	PointSet& s = c->sets[set_i];
	if (!s.zsort_ready) TNSX_FAIL(c, TNSX_ERR_STATE, "tns::TreeNSearch::apply_zsort error: no zsort order ready for set_i (%d).", set_i);
	if (host) { const tnsx_status r = zsort_host_order(c, s); if (r != TNSX_OK) return r; *host = s.zsort_host.data(); }
	if (dev) *dev = s.zsort_dev.as<int>();
	if (n) *n = s.zsort_n;
	return TNSX_OK;
}

tnsx_status tnsx_apply_zsort(tnsx_context* c, int set_i, void* data, size_t elem_bytes, int stride, int on_device)
{
	if (!c) return TNSX_ERR_INVALID;
	if (c->multi) {
		if (on_device) TNSX_FAIL(c, TNSX_ERR_STATE, "tnsx_apply_zsort on device memory: not available on a multi-device context");
		return tnsx_multi::apply_zsort(c->multi, set_i, data, elem_bytes, stride, c->last_error);
	}
	if (!set_ok(c, set_i)) TNSX_FAIL(c, TNSX_ERR_INVALID, "tns::TreeNSearch::apply_zsort error: set to z_sort does not exit.");
	PointSet& s = c->sets[set_i];
	if (!s.zsort_ready) TNSX_FAIL(c, TNSX_ERR_STATE, "tns::TreeNSearch::apply_zsort error: no zsort order ready for set_i (%d).", set_i);
	const int n = s.zsort_n;
	if (n == 0 || stride <= 0 || elem_bytes == 0) return TNSX_OK;
	if (!data) TNSX_FAIL(c, TNSX_ERR_INVALID, "apply_zsort: null data pointer");
	const size_t rec = elem_bytes * (size_t)stride;
	if (on_device) {
		if (hipSetDevice(c->device) != hipSuccess) TNSX_FAIL(c, TNSX_ERR_HIP, "hipSetDevice failed");
		HIPCHK(c, c->permute_tmp.reserve(rec * (size_t)n));
		HIPCHK(c, hipMemcpyAsync(c->permute_tmp.p, data, rec * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
		tnsx::launch_permute_bytes(c->permute_tmp.p, data, s.zsort_dev.as<int>(), n, rec, c->stream);
		HIPCHK(c, hipStreamSynchronize(c->stream));
	}
	else {
		// user memory on the host: gather through a swap buffer on the host cores, as TreeNSearch.h:456-480 does with OpenMP
		std::vector<unsigned char> swap((const unsigned char*)data, (const unsigned char*)data + rec * (size_t)n);
		unsigned char* dst = (unsigned char*)data;
		{ const tnsx_status r = zsort_host_order(c, s); if (r != TNSX_OK) return r; }
		const int* map = s.zsort_host.data();
		const unsigned char* src = swap.data();
		auto gather = [=](int lo, int hi) { for (int i = lo; i < hi; i++) std::memcpy(dst + rec * (size_t)i, src + rec * (size_t)map[i], rec); };
		unsigned n_thr = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
		if ((size_t)n * rec < ((size_t)1 << 20)) n_thr = 1;
		std::vector<std::thread> pool;
		for (unsigned t = 1; t < n_thr; t++) pool.emplace_back(gather, (int)((long long)n * t / n_thr), (int)((long long)n * (t + 1) / n_thr));
		gather(0, (int)((long long)n / n_thr));
		for (std::thread& th : pool) th.join();
	}
	return TNSX_OK;
}

tnsx_status tnsx_halo_pack(tnsx_context* c, const float* xyz, const float* radii, const long long* global_ids, int n_points, float left_cut,
                           float right_cut, float* out_left, float* out_right, unsigned long long capacity_left, unsigned long long capacity_right,
                           unsigned int* counts_dev, unsigned int* counts_host)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_NOT_MULTI("tnsx_halo_pack");
	if (n_points < 0 || !counts_dev || (n_points > 0 && (!xyz || !global_ids))) TNSX_FAIL(c, TNSX_ERR_INVALID, "tnsx_halo_pack: null pointer or negative size");
	if (hipSetDevice(c->device) != hipSuccess) TNSX_FAIL(c, TNSX_ERR_HIP, "hipSetDevice failed");
	HIPCHK(c, hipMemsetAsync(counts_dev, 0, 2 * sizeof(unsigned int), c->stream));
	tnsx::launch_halo_pack(xyz, radii, global_ids, n_points, left_cut, right_cut, out_left, out_right, capacity_left, capacity_right, counts_dev, c->stream);
	HIPCHK(c, hipGetLastError());
	if (counts_host) {
		HIPCHK(c, c->h_small.reserve(64));
		HIPCHK(c, hipMemcpyAsync(c->h_small.p, counts_dev, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		counts_host[0] = c->h_small.as<unsigned int>()[0];
		counts_host[1] = c->h_small.as<unsigned int>()[1];
	}
	return TNSX_OK;
}

tnsx_status tnsx_x_histogram(tnsx_context* c, const float* xyz, int n_points, float x0, float inv_dx, int n_bins, unsigned int* hist_dev)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_NOT_MULTI("tnsx_x_histogram");
	if (n_points < 0 || n_bins <= 0 || !hist_dev || (n_points > 0 && !xyz)) TNSX_FAIL(c, TNSX_ERR_INVALID, "tnsx_x_histogram: null pointer or bad size");
	if (hipSetDevice(c->device) != hipSuccess) TNSX_FAIL(c, TNSX_ERR_HIP, "hipSetDevice failed");
	tnsx::launch_x_histogram(xyz, n_points, x0, inv_dx, n_bins, hist_dev, c->stream);
	HIPCHK(c, hipGetLastError());
	return TNSX_OK;
}

tnsx_status tnsx_set_point_ids(tnsx_context* c, int set_i, const int* ids_dev)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_NOT_MULTI("tnsx_set_point_ids");
	if (!set_ok(c, set_i)) TNSX_FAIL(c, TNSX_ERR_INVALID, "tnsx_set_point_ids: set does not exist (%d)", set_i);
	c->sets[set_i].user_ids = ids_dev;
	return TNSX_OK;
}

tnsx_status tnsx_synchronize(tnsx_context* c)
{
	if (!c) return TNSX_ERR_INVALID;
	if (c->multi) return TNSX_OK;
	if (hipSetDevice(c->device) != hipSuccess) TNSX_FAIL(c, TNSX_ERR_HIP, "hipSetDevice failed");
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return TNSX_OK;
}

tnsx_status tnsx_set_query_count(tnsx_context* c, int set_i, int n_query)
{
	if (!c) return TNSX_ERR_INVALID;
	TNSX_NOT_MULTI("tnsx_set_query_count");
	if (!set_ok(c, set_i)) TNSX_FAIL(c, TNSX_ERR_INVALID, "tnsx_set_query_count: set does not exist (%d)", set_i);
	c->sets[set_i].n_query = n_query < 0 ? -1 : n_query;
	return TNSX_OK;
}

// (tnsx_slab.cpp) the stream / device a single-device context works on
void* tnsx_internal_stream(tnsx_context* c) { return c && !c->multi ? (void*)c->stream : nullptr; }
int tnsx_internal_device(tnsx_context* c) { return c ? c->device : 0; }
extern "C" int tnsx_get_device(const tnsx_context* c) { return c ? c->device : -1; }
void tnsx_internal_set_sync_timeout(tnsx_context* c, double seconds) { if (c) c->sync_timeout_s = seconds; }

tnsx_status tnsx_get_stats(const tnsx_context* c, tnsx_stats* out)
{
	if (!c || !out) return TNSX_ERR_INVALID;
	if (c->multi) { tnsx_multi::stats(c->multi, out); return TNSX_OK; }
	*out = c->stats;
	out->zsort_cell_size_inv = c->zsort_inv_h;
	for (int d = 0; d < 3; d++) { out->world_bottom[d] = c->world[d]; out->world_top[d] = c->world[3 + d]; }   // (prepare_zsort may have moved the box)
	out->world_cells_pow2 = c->world_cells_pow2;
	return TNSX_OK;
}

}  // extern "C"
