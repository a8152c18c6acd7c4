// Multi-device mode of the C ABI: see tnsx_multi.h.  Host-side orchestration only -- every device runs the ordinary single-device
// engine (through the C ABI itself) on its slab [owned | ghosts]; there is no CPU search path in here either.
#include "tnsx_multi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <thread>

namespace tnsx_multi {
namespace {

// ---------------------------------------------------------------------------------------------------------------- helpers
unsigned host_threads()
{
	return std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
}
// f(chunk_index, begin, end) over [0, n) cut into `n_chunks` equal pieces, on up to host_threads() threads
void parallel_chunks(size_t n, size_t n_chunks, const std::function<void(size_t, size_t, size_t)>& f)
{
	if (n_chunks == 0) return;
	const unsigned T = (unsigned)std::min<size_t>(host_threads(), n_chunks);
	auto work = [&](unsigned t) {
		for (size_t c = t; c < n_chunks; c += T) f(c, n * c / n_chunks, n * (c + 1) / n_chunks);
	};
	std::vector<std::thread> pool;
	for (unsigned t = 1; t < T; t++) pool.emplace_back(work, t);
	work(0);
	for (std::thread& th : pool) th.join();
}

struct Pinned {
	void* p = nullptr;
	size_t cap = 0;
	~Pinned() { if (p) (void)hipHostFree(p); }
	Pinned() = default;
	Pinned(const Pinned&) = delete;
	Pinned& operator=(const Pinned&) = delete;
	Pinned(Pinned&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
	bool reserve(size_t bytes)
	{
		if (bytes <= cap && p) return true;
		if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
		const size_t want = bytes + bytes / 8 + 256;
		if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) { p = nullptr; return false; }   // portable: every device may DMA to / from it
		cap = want;
		return true;
	}
	template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct HostSet {
	const void* xyz = nullptr;
	const void* radii = nullptr;
	int n = 0;
	bool is_double = false, has_radii = false;
	bool radii_is_double = false;        // element type of `radii`: set where radii are handed over, kept by a points-only resize
	                                     // (set_radii / set_radii_double of the reference are separate arrays, TreeNSearch.h:375-378)
	std::vector<float> f32_xyz, f32_r;   // (float) casts of double inputs, TreeNSearch.cpp:277-296
	const float* x = nullptr;            // what this run reads
	const float* r = nullptr;
};

struct SlabSet {   // one user set as one slab sees it: [owned | ghosts]
	Pinned xyz, radii, gid_buf;          // pinned: uploaded by DMA every run
	int* gid = nullptr;                  // global ids of [owned | ghosts] (= gid_buf)
	int n_owned = 0, n_ghost = 0;
	int* d_ids = nullptr;
	size_t d_ids_cap = 0;
};

struct PairLocal { uint64_t n_records = 0, n_neighbors = 0; int n_points = 0; };

struct Device {
	int id = 0;
	tnsx_context* ctx = nullptr;
	std::vector<SlabSet> sets;
	std::vector<PairLocal> pairs;   // [i * n_sets + j]
	Pinned local_offsets;
	tnsx_stats st{};
	std::string error;
	tnsx_status status = TNSX_OK;
};

struct PairOut {
	bool valid = false;
	int n_i = 0;
	uint64_t n_records = 0, n_neighbors = 0;
	Pinned offsets, records;
};

}  // namespace

struct State {
	tnsx_options opt{};
	std::vector<Device> dev;
	std::vector<HostSet> sets;
	std::vector<std::vector<char>> active;
	std::vector<PairOut> pairs;
	int n_sets_at_last_run = 0;
	bool ran = false;
	bool symmetric = true, radius_set = false;
	float radius = -1.0f, cell_size = -1.0f;
	int n_sets_with_radii = 0;
	int arith = TNSX_ARITH_STRICT;
	tnsx_context* zctx = nullptr;   // full-set engine on the first device: prepare_zsort / apply_zsort (made on first use)
	int zctx_sets = 0;
	tnsx_stats stats{};
	int n_slabs_last = 0;
	// the cuts of the previous run are kept while nothing they depend on has changed (any cuts at least one halo apart are
	// CORRECT; only the balance drifts with the points) and are re-balanced from a fresh histogram every CUT_PERIOD runs
	std::vector<float> cuts;
	float cuts_halo = 0.0f;
	int64_t cuts_n_total = -1;
	int cuts_age = 0;
};

namespace {
#define MFAIL(code, ...)                                    \
	do {                                                    \
		char _b[512];                                       \
		std::snprintf(_b, sizeof(_b), __VA_ARGS__);         \
		error = _b;                                         \
		return (code);                                      \
	} while (0)

bool set_ok(const State* m, int s) { return s >= 0 && s < (int)m->sets.size(); }

tnsx_status ensure_zctx(State* m, std::string& error)
{
	if (!m->zctx) {
		tnsx_options o = m->opt;
		o.n_devices = 0;
		o.device_id = m->dev[0].id;
		o.mirror_to_host = 0;
		if (tnsx_create(&o, &m->zctx) != TNSX_OK) MFAIL(TNSX_ERR_HIP, "multi-device mode: %s", tnsx_last_error(nullptr));
		m->zctx_sets = 0;
	}
	if (m->radius_set) (void)tnsx_set_search_radius(m->zctx, m->radius);
	if (m->cell_size > 0.0f) (void)tnsx_set_cell_size(m->zctx, m->cell_size);   // (write-once in the engine: later calls fail silently, same value)
	for (int s = 0; s < (int)m->sets.size(); s++) {
		const HostSet& h = m->sets[(size_t)s];
		const unsigned common = TNSX_HOST | (h.has_radii ? TNSX_VARIABLE : 0u);
		const unsigned flags_r = ((h.has_radii ? h.radii_is_double : h.is_double) ? TNSX_F64 : TNSX_F32) | common;
		const unsigned flags_x = (h.is_double ? TNSX_F64 : TNSX_F32) | common;
		// registered with the element type of the radii; points of another type follow as a points-only resize (pointers are only
		// stored here, they are read at the run)
		if (s >= m->zctx_sets) {
			if (tnsx_add_point_set(m->zctx, h.xyz, h.radii, h.n, flags_r) < 0) MFAIL(TNSX_ERR_INVALID, "%s", tnsx_last_error(m->zctx));
			m->zctx_sets++;
		}
		else if (tnsx_resize_point_set(m->zctx, s, h.xyz, h.radii, h.n, flags_r) != TNSX_OK) MFAIL(TNSX_ERR_INVALID, "%s", tnsx_last_error(m->zctx));
		if (flags_x != flags_r && tnsx_resize_point_set(m->zctx, s, h.xyz, nullptr, h.n, flags_x) != TNSX_OK) MFAIL(TNSX_ERR_INVALID, "%s", tnsx_last_error(m->zctx));
	}
	return TNSX_OK;
}
}  // namespace

// ================================================================================================================ lifetime
State* create(const tnsx_options& opt, std::string& error)
{
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { error = "tnsx_create: no HIP device available (this engine has no CPU fallback)"; return nullptr; }
	State* m = new State();
	m->opt = opt;
	m->arith = opt.arith;
	const int want = std::min(opt.n_devices, (int)(sizeof(opt.device_ids) / sizeof(opt.device_ids[0])));
	for (int k = 0; k < want; k++) {
		Device d;
		d.id = opt.device_ids[k];
		if (d.id < 0 || d.id >= n_dev) { error = "tnsx_create: device id out of range in tnsx_options.device_ids"; destroy(m); return nullptr; }
		tnsx_options o = opt;
		o.n_devices = 0;
		o.device_id = d.id;
		o.stream = nullptr;          // every engine makes its own stream on its own device
		o.mirror_to_host = 0;        // the lists of all devices are gathered into ONE pinned buffer by this layer
		if (tnsx_create(&o, &d.ctx) != TNSX_OK) { error = std::string("tnsx_create (multi-device): ") + tnsx_last_error(nullptr); destroy(m); return nullptr; }
		m->dev.push_back(std::move(d));
	}
	return m;
}

void destroy(State* m)
{
	if (!m) return;
	for (Device& d : m->dev) {
		(void)hipSetDevice(d.id);
		for (SlabSet& s : d.sets) if (s.d_ids) (void)hipFree(s.d_ids);
		if (d.ctx) tnsx_destroy(d.ctx);
	}
	if (m->zctx) tnsx_destroy(m->zctx);
	delete m;
}

// ================================================================================================================ configuration
int add_point_set(State* m, const void* xyz, const void* radii, int n, unsigned flags, std::string& error)
{
	if (n < 0) { error = "add_point_set: n_points < 0"; return -TNSX_ERR_INVALID; }
	if (flags & TNSX_DEVICE) { error = "multi-device mode takes host pointers only (device-resident data: one process per GPU, treensearch_amd/multi.py)"; return -TNSX_ERR_INVALID; }
	HostSet h;
	h.xyz = xyz; h.radii = radii; h.n = n;
	h.is_double = (flags & TNSX_F64) != 0;
	h.radii_is_double = h.is_double;
	h.has_radii = radii != nullptr || (flags & TNSX_VARIABLE);
	if (h.has_radii) m->n_sets_with_radii++;
	m->sets.push_back(std::move(h));
	const size_t ns = m->sets.size();
	for (auto& row : m->active) row.push_back(0);
	m->active.emplace_back(ns, 0);
	return (int)ns - 1;
}

tnsx_status resize_point_set(State* m, int set_id, const void* xyz, const void* radii, int n, unsigned flags, std::string& error)
{
	if (!set_ok(m, set_id)) MFAIL(TNSX_ERR_INVALID, "TreeNSearch::resize_point_set error: Cannot resize a set that was not previously added.");
	if (n < 0) MFAIL(TNSX_ERR_INVALID, "resize_point_set: n_points < 0");
	if (flags & TNSX_DEVICE) MFAIL(TNSX_ERR_INVALID, "multi-device mode takes host pointers only");
	const bool with_radii = radii != nullptr || (flags & TNSX_VARIABLE);
	if (with_radii && m->n_sets_with_radii == 0) MFAIL(TNSX_ERR_INVALID, "TreeNSearch::resize_point_set error: Cannot resize a set with a radii array if it previously didn't have one.");
	HostSet& h = m->sets[(size_t)set_id];
	h.xyz = xyz; h.n = n;
	h.is_double = (flags & TNSX_F64) != 0;
	if (with_radii) { h.radii = radii; h.radii_is_double = h.is_double; }
	return TNSX_OK;
}

tnsx_status set_search_radius(State* m, float r, std::string& error)
{
	if (m->n_sets_with_radii > 0) MFAIL(TNSX_ERR_INVALID, "tns::TreeNSearch::set_search_radius error: Cannot set a global search radius if a set with a radii array was already added.");
	m->radius_set = true;
	m->radius = r;
	return TNSX_OK;
}
tnsx_status set_cell_size(State* m, float cell, std::string& error)
{
	if (m->cell_size > 0.0f) MFAIL(TNSX_ERR_INVALID, "tns::TreeNSearch::set_cell_size error: Cell size already set. Create a new TreeNSearch instance if you need a different cell_size.");
	m->cell_size = cell;
	return TNSX_OK;
}
void set_symmetric(State* m, bool on) { m->symmetric = on; }
void set_arithmetic(State* m, int arith) { m->arith = arith; }
tnsx_status set_active(State* m, int i, int j, bool on, std::string& error)
{
	if (!set_ok(m, i) || !set_ok(m, j)) MFAIL(TNSX_ERR_INVALID, "set_active_search: set does not exist (%d, %d)", i, j);
	m->active[(size_t)i][(size_t)j] = on;
	return TNSX_OK;
}
tnsx_status set_active_all(State* m, int i, bool search_in_all, bool be_found_by_all, std::string& error)
{
	if (!set_ok(m, i)) MFAIL(TNSX_ERR_INVALID, "set_active_search: set does not exist (%d)", i);
	for (size_t j = 0; j < m->sets.size(); j++) m->active[j][(size_t)i] = be_found_by_all;   // column first, then row (TreeNSearch.cpp:223-232)
	for (size_t j = 0; j < m->sets.size(); j++) m->active[(size_t)i][j] = search_in_all;
	return TNSX_OK;
}
void set_all_searches(State* m, bool on) { for (auto& row : m->active) for (auto& v : row) v = on; }
int n_sets(const State* m) { return (int)m->sets.size(); }
int n_points_in_set(const State* m, int s) { return set_ok(m, s) ? m->sets[(size_t)s].n : -1; }
int64_t total_points(const State* m) { int64_t t = 0; for (const HostSet& h : m->sets) t += h.n; return t; }
bool is_active(const State* m, int i, int j) { return set_ok(m, i) && set_ok(m, j) && m->active[(size_t)i][(size_t)j]; }
uint64_t neighborlist_bytes(const State* m) { uint64_t b = 0; for (const PairOut& p : m->pairs) if (p.valid) b += p.n_records * sizeof(int); return b; }

// ================================================================================================================ run
tnsx_status run(State* m, std::string& error)
{
	const int S = (int)m->sets.size();
	const int D = (int)m->dev.size();
	m->ran = false;
	// ---- _check (TreeNSearch.cpp:366-392)
	if (m->radius_set && m->radius <= 0.0f) MFAIL(TNSX_ERR_CONFIG, "TreeNSearch error: global_search_radius <= 0.");
	if (m->radius_set && m->n_sets_with_radii > 0) MFAIL(TNSX_ERR_CONFIG, "TreeNSearch error: global search radius and per-point variable search radii specified.");
	if (!m->radius_set && m->n_sets_with_radii != S) MFAIL(TNSX_ERR_CONFIG, "TreeNSearch error: not all point sets have per-point search radius specified.");
	const bool variable = !m->radius_set;

	// ---- inputs as floats (the reference re-reads the user's pointers at every run, TreeNSearch.h:375-378)
	int64_t n_total = 0;
	for (HostSet& h : m->sets) {
		h.x = nullptr; h.r = nullptr;
		if (h.n == 0) continue;
		if (!h.xyz || (h.has_radii && !h.radii)) MFAIL(TNSX_ERR_INVALID, "point set with n > 0 has a null pointer");
		n_total += h.n;
		if (h.is_double) {
			h.f32_xyz.resize(3 * (size_t)h.n);
			const double* src = (const double*)h.xyz;
			float* dst = h.f32_xyz.data();
			parallel_chunks(3 * (size_t)h.n, 64, [&](size_t, size_t b, size_t e) { for (size_t i = b; i < e; i++) dst[i] = (float)src[i]; });
			h.x = dst;
		}
		else h.x = (const float*)h.xyz;
		if (h.has_radii && h.radii_is_double) {
			h.f32_r.resize((size_t)h.n);
			const double* rs = (const double*)h.radii;
			float* rd = h.f32_r.data();
			parallel_chunks((size_t)h.n, 64, [&](size_t, size_t b, size_t e) { for (size_t i = b; i < e; i++) rd[i] = (float)rs[i]; });
			h.r = rd;
		}
		else h.r = h.has_radii ? (const float*)h.radii : nullptr;
	}

	// ---- x range and largest radius
	float x0 = FLT_MAX, x1 = -FLT_MAX, r_max = variable ? 0.0f : m->radius;
	{
		const size_t NC = 64;
		std::vector<float> lo(NC * (size_t)std::max(S, 1), FLT_MAX), hi(NC * (size_t)std::max(S, 1), -FLT_MAX), rm(NC * (size_t)std::max(S, 1), 0.0f);
		for (int s = 0; s < S; s++) {
			const HostSet& h = m->sets[(size_t)s];
			if (h.n == 0) continue;
			parallel_chunks((size_t)h.n, NC, [&](size_t c, size_t b, size_t e) {
				float l = FLT_MAX, u = -FLT_MAX, r = 0.0f;
				for (size_t i = b; i < e; i++) { const float x = h.x[3 * i]; if (x == x) { l = std::min(l, x); u = std::max(u, x); } if (variable) r = std::max(r, h.r[i]); }
				lo[(size_t)s * NC + c] = l; hi[(size_t)s * NC + c] = u; rm[(size_t)s * NC + c] = r;
			});
		}
		for (float v : lo) x0 = std::min(x0, v);
		for (float v : hi) x1 = std::max(x1, v);
		if (variable) for (float v : rm) r_max = std::max(r_max, v);
	}
	if (n_total > 0 && (!(r_max > 0.0f) || !std::isfinite(r_max))) MFAIL(TNSX_ERR_CONFIG, "TreeNSearch error: search radius must be > 0");
	if (n_total > 0 && !(std::isfinite(x0) && std::isfinite(x1))) MFAIL(TNSX_ERR_INVALID, "a point coordinate is not finite");

	// ---- balanced cuts at plane granularity (treensearch_amd/multi.py: SlabDecomposition._cuts); one plane >= one halo width
	const float halo = r_max * 1.001f;
	int K = 1;                             // slabs in use
	std::vector<float> cuts(2, 0.0f);      // cuts[k] <= x < cuts[k+1]; cuts[0] = -inf, cuts[K] = +inf
	constexpr int CUT_PERIOD = 16;
	if (n_total > 0 && m->cuts_n_total == n_total && m->cuts_halo == halo && m->cuts_age < CUT_PERIOD && !m->cuts.empty()) {
		cuts = m->cuts;
		K = (int)cuts.size() - 1;
		m->cuts_age++;
	}
	else if (n_total > 0) {
		double w = (double)halo * 1.001;
		const double ext = (double)x1 - (double)x0;
		if (ext / w > 1048576.0) w = ext / 1048576.0;
		const int n_planes = (int)(ext / w) + 1;
		K = std::max(1, std::min(D, n_planes));
		const float wf = (float)w, inv = 1.0f / wf;
		const size_t NC = 64;
		std::vector<uint64_t> hist((size_t)n_planes, 0);
		{
			std::vector<std::vector<uint32_t>> part(NC);
			for (int s = 0; s < S; s++) {
				const HostSet& h = m->sets[(size_t)s];
				if (h.n == 0) continue;
				for (auto& p : part) p.assign((size_t)n_planes, 0);
				parallel_chunks((size_t)h.n, NC, [&](size_t c, size_t b, size_t e) {
					std::vector<uint32_t>& p = part[c];
					for (size_t i = b; i < e; i++) {
						const float x = h.x[3 * i];
						if (x != x) continue;
						int pl = (int)((x - x0) * inv);
						pl = pl < 0 ? 0 : (pl > n_planes - 1 ? n_planes - 1 : pl);
						p[(size_t)pl]++;
					}
				});
				for (const auto& p : part) for (int b = 0; b < n_planes; b++) hist[(size_t)b] += p[(size_t)b];
			}
		}
		std::vector<uint64_t> cum((size_t)n_planes);
		uint64_t acc = 0;
		for (int b = 0; b < n_planes; b++) { acc += hist[(size_t)b]; cum[(size_t)b] = acc; }
		cuts.assign((size_t)K + 1, 0.0f);
		cuts[0] = -INFINITY; cuts[(size_t)K] = INFINITY;
		int prev = 0;
		for (int k = 1; k < K; k++) {
			const double target = (double)acc * k / K;
			int b = (int)(std::lower_bound(cum.begin(), cum.end(), (uint64_t)std::ceil(target)) - cum.begin()) + 1;
			if (b >= 2 && std::fabs((double)cum[(size_t)b - 2] - target) <= std::fabs((double)cum[(size_t)std::min(b, n_planes) - 1] - target)) b -= 1;
			b = std::min(std::max(b, prev + 1), n_planes - (K - k));
			cuts[(size_t)k] = x0 + (float)b * wf;
			prev = b;
		}
		m->cuts = cuts; m->cuts_halo = halo; m->cuts_n_total = n_total; m->cuts_age = 0;
	}
	else { cuts[0] = -INFINITY; cuts[1] = INFINITY; }
	m->n_slabs_last = K;

	// ---- partition every set: owned points of slab k in original order, then the ghosts (points of slab k -+ 1 within one halo
	//      width of the shared face), also in original order -- deterministic whatever the thread count
	for (Device& d : m->dev) { d.sets.resize((size_t)S); d.status = TNSX_OK; d.error.clear(); }
	for (int s = 0; s < S; s++) {
		const HostSet& h = m->sets[(size_t)s];
		const size_t NC = h.n > 0 ? std::min<size_t>(256, ((size_t)h.n + 65535) / 65536 * 4) : 0;
		// per chunk and slab: owned count, ghost count
		std::vector<uint32_t> cnt(std::max<size_t>(NC, 1) * (size_t)D * 2, 0);
		auto owner_of = [&](float x) { int k = 0; while (k + 1 < K && x >= cuts[(size_t)k + 1]) k++; return k; };
		if (h.n > 0) {
			parallel_chunks((size_t)h.n, NC, [&](size_t c, size_t b, size_t e) {
				uint32_t* my = cnt.data() + c * (size_t)D * 2;
				for (size_t i = b; i < e; i++) {
					const float x = h.x[3 * i];
					if (x != x) continue;                                         // NaN x: no point
					const int k = owner_of(x);
					my[2 * k]++;
					if (k > 0 && x < cuts[(size_t)k] + halo) my[2 * (k - 1) + 1]++;          // ghost of the left neighbour
					if (k + 1 < K && x >= cuts[(size_t)k + 1] - halo) my[2 * (k + 1) + 1]++;  // ghost of the right neighbour
				}
			});
		}
		// exclusive prefix over the chunks -> where every chunk writes
		std::vector<uint32_t> n_owned((size_t)D, 0), n_ghost((size_t)D, 0);
		for (size_t c = 0; c < NC; c++) {
			for (int k = 0; k < D; k++) {
				uint32_t* my = cnt.data() + c * (size_t)D * 2;
				const uint32_t o = my[2 * k], g = my[2 * k + 1];
				my[2 * k] = n_owned[(size_t)k]; my[2 * k + 1] = n_ghost[(size_t)k];
				n_owned[(size_t)k] += o; n_ghost[(size_t)k] += g;
			}
		}
		for (int k = 0; k < D; k++) {
			SlabSet& ss = m->dev[(size_t)k].sets[(size_t)s];
			ss.n_owned = (int)n_owned[(size_t)k]; ss.n_ghost = (int)n_ghost[(size_t)k];
			const size_t tot = (size_t)ss.n_owned + ss.n_ghost;
			if (!ss.xyz.reserve(std::max<size_t>(tot, 1) * 3 * sizeof(float)) || !ss.gid_buf.reserve(std::max<size_t>(tot, 1) * sizeof(int)) ||
			    (h.has_radii && !ss.radii.reserve(std::max<size_t>(tot, 1) * sizeof(float))))
				MFAIL(TNSX_ERR_HIP, "multi-device mode: pinned host memory exhausted");
			ss.gid = ss.gid_buf.as<int>();
		}
		if (h.n > 0) {
			parallel_chunks((size_t)h.n, NC, [&](size_t c, size_t b, size_t e) {
				std::vector<uint32_t> pos(cnt.begin() + (long)(c * (size_t)D * 2), cnt.begin() + (long)((c + 1) * (size_t)D * 2));
				auto put = [&](int k, bool ghost, size_t i) {
					SlabSet& ss = m->dev[(size_t)k].sets[(size_t)s];
					const size_t p = ghost ? (size_t)ss.n_owned + pos[(size_t)(2 * k + 1)]++ : pos[(size_t)(2 * k)]++;
					float* o = ss.xyz.as<float>() + 3 * p;
					o[0] = h.x[3 * i]; o[1] = h.x[3 * i + 1]; o[2] = h.x[3 * i + 2];
					if (h.has_radii) ss.radii.as<float>()[p] = h.r[i];
					ss.gid[p] = (int)i;
				};
				for (size_t i = b; i < e; i++) {
					const float x = h.x[3 * i];
					if (x != x) continue;
					const int k = owner_of(x);
					put(k, false, i);
					if (k > 0 && x < cuts[(size_t)k] + halo) put(k - 1, true, i);
					if (k + 1 < K && x >= cuts[(size_t)k + 1] - halo) put(k + 1, true, i);
				}
			});
		}
	}

	// ---- every device: upload its slab, run, report the sizes of its lists
	struct Job { int i, j; };
	std::vector<Job> jobs;
	for (int i = 0; i < S; i++) for (int j = 0; j < S; j++) if (m->active[(size_t)i][(size_t)j]) jobs.push_back({ i, j });
	auto device_run = [&](int k) {
		Device& d = m->dev[(size_t)k];
		auto fail = [&](tnsx_status st, const char* what) { d.status = st; d.error = std::string(what) + ": " + tnsx_last_error(d.ctx); };
		if (hipSetDevice(d.id) != hipSuccess) { d.status = TNSX_ERR_HIP; d.error = "hipSetDevice failed"; return; }
		tnsx_context* c = d.ctx;
		if (m->radius_set && tnsx_set_search_radius(c, m->radius) != TNSX_OK) return fail(TNSX_ERR_CONFIG, "set_search_radius");
		(void)tnsx_set_symmetric_search(c, m->symmetric ? 1 : 0);
		(void)tnsx_set_arithmetic(c, m->arith);
		if (m->cell_size > 0.0f) (void)tnsx_set_cell_size(c, m->cell_size);   // (write-once in the engine: later calls fail, same value)
		for (int s = 0; s < S; s++) {
			SlabSet& ss = d.sets[(size_t)s];
			const HostSet& h = m->sets[(size_t)s];
			const int n = ss.n_owned + ss.n_ghost;
			const unsigned flags = TNSX_F32 | TNSX_HOST | (h.has_radii ? TNSX_VARIABLE : 0u);
			const void* rp = h.has_radii && n > 0 ? ss.radii.p : nullptr;
			if (s >= tnsx_get_n_sets(c)) { if (tnsx_add_point_set(c, n > 0 ? ss.xyz.p : nullptr, rp, n, flags) < 0) return fail(TNSX_ERR_INVALID, "add_point_set"); }
			else if (tnsx_resize_point_set(c, s, n > 0 ? ss.xyz.p : nullptr, rp, n, flags) != TNSX_OK) return fail(TNSX_ERR_INVALID, "resize_point_set");
			(void)tnsx_set_query_count(c, s, ss.n_owned);
			if (n > 0) {
				if ((size_t)n > ss.d_ids_cap) {
					if (ss.d_ids) (void)hipFree(ss.d_ids);
					ss.d_ids = nullptr; ss.d_ids_cap = 0;
					const size_t want = (size_t)n + (size_t)n / 8 + 64;
					if (hipMalloc((void**)&ss.d_ids, want * sizeof(int)) != hipSuccess) { d.status = TNSX_ERR_HIP; d.error = "hipMalloc of the id array failed"; return; }
					ss.d_ids_cap = want;
				}
				if (hipMemcpy(ss.d_ids, ss.gid, (size_t)n * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { d.status = TNSX_ERR_HIP; d.error = "upload of the id array failed"; return; }
			}
			(void)tnsx_set_point_ids(c, s, n > 0 ? ss.d_ids : nullptr);
		}
		for (int i = 0; i < S; i++) for (int j = 0; j < S; j++) (void)tnsx_set_active_search(c, i, j, m->active[(size_t)i][(size_t)j] ? 1 : 0);
		const tnsx_status r = tnsx_run(c);
		if (r != TNSX_OK) return fail(r, "run");
		d.pairs.assign((size_t)S * S, PairLocal());
		for (const Job& jb : jobs) {
			tnsx_csr_view v;
			if (tnsx_get_pair_view(c, jb.i, jb.j, &v) != TNSX_OK) return fail(TNSX_ERR_STATE, "get_pair_view");
			PairLocal& pl = d.pairs[(size_t)jb.i * S + jb.j];
			pl.n_records = v.n_records; pl.n_neighbors = v.n_neighbors; pl.n_points = v.n_points;
		}
		(void)tnsx_get_stats(c, &d.st);
	};
	{
		std::vector<std::thread> pool;
		for (int k = 1; k < D; k++) pool.emplace_back(device_run, k);
		device_run(0);
		for (std::thread& th : pool) th.join();
	}
	for (const Device& d : m->dev) if (d.status != TNSX_OK) { error = "device " + std::to_string(d.id) + ": " + d.error; return d.status; }

	// ---- one pinned buffer per pair for the records of all devices; offsets by ORIGINAL point index into it
	m->pairs.resize((size_t)S * S);
	for (PairOut& p : m->pairs) p.valid = false;
	m->n_sets_at_last_run = S;
	std::vector<std::vector<uint64_t>> base(jobs.size(), std::vector<uint64_t>((size_t)D + 1, 0));
	for (size_t q = 0; q < jobs.size(); q++) {
		const Job& jb = jobs[q];
		PairOut& po = m->pairs[(size_t)jb.i * S + jb.j];
		po.n_i = m->sets[(size_t)jb.i].n;
		po.n_neighbors = 0;
		// int 0 of the gathered records is a shared EMPTY record and every offset starts out pointing at it: a point no slab owns
		// (x = NaN: no point) then has an empty list instead of an offset nobody wrote
		base[q][0] = 1;
		for (int k = 0; k < D; k++) {
			const PairLocal& pl = m->dev[(size_t)k].pairs[(size_t)jb.i * S + jb.j];
			base[q][(size_t)k + 1] = base[q][(size_t)k] + pl.n_records;
			po.n_neighbors += pl.n_neighbors;
		}
		po.n_records = base[q][(size_t)D];
		if (!po.records.reserve(std::max<uint64_t>(po.n_records, 1) * sizeof(int)) || !po.offsets.reserve((size_t)std::max(po.n_i, 1) * sizeof(uint64_t)))
			MFAIL(TNSX_ERR_HIP, "multi-device mode: pinned host memory exhausted (lists: %llu ints)", (unsigned long long)po.n_records);
		po.records.as<int>()[0] = 0;
		{
			uint64_t* go = po.offsets.as<uint64_t>();
			parallel_chunks((size_t)std::max(po.n_i, 1), 8, [&](size_t, size_t b, size_t e) { std::memset(go + b, 0, (e - b) * sizeof(uint64_t)); });
		}
	}
	auto device_fetch = [&](int k) {
		Device& d = m->dev[(size_t)k];
		if (hipSetDevice(d.id) != hipSuccess) { d.status = TNSX_ERR_HIP; d.error = "hipSetDevice failed"; return; }
		for (size_t q = 0; q < jobs.size(); q++) {
			const Job& jb = jobs[q];
			PairOut& po = m->pairs[(size_t)jb.i * S + jb.j];
			const SlabSet& ss = d.sets[(size_t)jb.i];
			const PairLocal& pl = d.pairs[(size_t)jb.i * S + jb.j];
			if (pl.n_points == 0) continue;
			if (!d.local_offsets.reserve((size_t)pl.n_points * sizeof(uint64_t))) { d.status = TNSX_ERR_HIP; d.error = "pinned host memory exhausted"; return; }
			if (tnsx_copy_pair(d.ctx, jb.i, jb.j, d.local_offsets.as<uint64_t>(), po.records.as<int>() + base[q][(size_t)k], 0) != TNSX_OK) {
				d.status = TNSX_ERR_HIP; d.error = std::string("copy_pair: ") + tnsx_last_error(d.ctx); return;
			}
			const uint64_t* lo = d.local_offsets.as<uint64_t>();
			uint64_t* go = po.offsets.as<uint64_t>();
			const uint64_t b0 = base[q][(size_t)k];
			const int* gid = ss.gid;
			parallel_chunks((size_t)pl.n_points, 8, [&](size_t, size_t b, size_t e) { for (size_t p = b; p < e; p++) go[(size_t)gid[p]] = b0 + lo[p]; });
		}
	};
	{
		std::vector<std::thread> pool;
		for (int k = 1; k < D; k++) pool.emplace_back(device_fetch, k);
		device_fetch(0);
		for (std::thread& th : pool) th.join();
	}
	for (const Device& d : m->dev) if (d.status != TNSX_OK) { error = "device " + std::to_string(d.id) + ": " + d.error; return d.status; }
	for (const Job& jb : jobs) m->pairs[(size_t)jb.i * S + jb.j].valid = true;

	// ---- statistics: sums over the devices, times = the slowest device
	tnsx_stats& T = m->stats;
	std::memset(&T, 0, sizeof(T));
	T.n_sets = S;
	T.n_points = (uint64_t)n_total;
	for (const Device& d : m->dev) {
		T.n_queries += d.st.n_queries; T.n_neighbors += d.st.n_neighbors; T.n_occupied_cells += d.st.n_occupied_cells;
		T.bytes_build += d.st.bytes_build; T.bytes_query += d.st.bytes_query;
		T.pool_retries += d.st.pool_retries; T.cold_passes += d.st.cold_passes; T.speculation_redos += d.st.speculation_redos;
		T.n_pool_pairs = std::max(T.n_pool_pairs, d.st.n_pool_pairs);
		T.ms_total = std::max(T.ms_total, d.st.ms_total); T.ms_fill = std::max(T.ms_fill, d.st.ms_fill); T.ms_sort = std::max(T.ms_sort, d.st.ms_sort);
	}
	T.n_devices_used = K;
	m->ran = true;
	return TNSX_OK;
}

// ================================================================================================================ results
tnsx_status pair_view(State* m, int i, int j, tnsx_csr_view* out, std::string& error)
{
	if (!m->ran) MFAIL(TNSX_ERR_STATE, "neighbour lists requested before a successful run()");
	const int n = m->n_sets_at_last_run;
	if (i < 0 || j < 0 || i >= n || j >= n) MFAIL(TNSX_ERR_INVALID, "TreeNSearch::get_neighborlist error: Set does not exist.");
	PairOut& p = m->pairs[(size_t)i * n + j];
	if (!p.valid) MFAIL(TNSX_ERR_STATE, "TreeNSearch::get_neighborlist error: Set pair not active.");
	out->n_points = p.n_i;
	out->n_records = p.n_records;
	out->n_neighbors = p.n_neighbors;
	out->offsets_device = nullptr;        // the lists live on several devices: host views only
	out->records_device = nullptr;
	out->offsets_host = p.offsets.as<uint64_t>();
	out->records_host = p.records.as<int>();
	return TNSX_OK;
}

// ================================================================================================================ zsort
tnsx_status prepare_zsort(State* m, std::string& error)
{
	{ const tnsx_status r = ensure_zctx(m, error); if (r != TNSX_OK) return r; }
	const tnsx_status r = tnsx_prepare_zsort(m->zctx);
	if (r != TNSX_OK) error = tnsx_last_error(m->zctx);
	return r;
}
tnsx_status zsort_order(State* m, int set_i, const int** host, int* n, std::string& error)
{
	if (!m->zctx) MFAIL(TNSX_ERR_STATE, "tns::TreeNSearch::apply_zsort error: no zsort order ready for set_i (%d).", set_i);
	const tnsx_status r = tnsx_get_zsort_order(m->zctx, set_i, host, nullptr, n);
	if (r != TNSX_OK) error = tnsx_last_error(m->zctx);
	return r;
}
tnsx_status apply_zsort(State* m, int set_i, void* data, size_t elem_bytes, int stride, std::string& error)
{
	if (!m->zctx) MFAIL(TNSX_ERR_STATE, "tns::TreeNSearch::apply_zsort error: no zsort order ready for set_i (%d).", set_i);
	const tnsx_status r = tnsx_apply_zsort(m->zctx, set_i, data, elem_bytes, stride, 0);
	if (r != TNSX_OK) error = tnsx_last_error(m->zctx);
	return r;
}

void stats(const State* m, tnsx_stats* out)
{
	*out = m->stats;
	if (m->zctx) {
		tnsx_stats z;
		if (tnsx_get_stats(m->zctx, &z) == TNSX_OK) {
			for (int d = 0; d < 3; d++) { out->world_bottom[d] = z.world_bottom[d]; out->world_top[d] = z.world_top[d]; }
			out->world_cells_pow2 = z.world_cells_pow2;
		}
	}
}

}  // namespace tnsx_multi
