// tns::TreeNSearch -- header-only drop-in for the reference class of the same name
// (InteractiveComputerGraphics/TreeNSearch, TreeNSearch/source/TreeNSearch.h:28-427), implemented on top of the
// MI355X engine's C ABI (include/tnsx.h, libtnsx.so).  Existing callers (SPlisHSPlasH, the reference's own
// tests) compile against this header unchanged: same method names, overloads, defaults and error behaviour
// (message on std::cout, then exit(-1)).
//
// Data flow: the user's raw pointers are re-read at every run() (host -> HBM), the search runs on the GPU, and
// the neighbour records are mirrored into pinned host memory before run() returns, so get_neighborlist() stays
// an O(1), lock-free, thread-safe read exactly like the reference's (TreeNSearch.cpp:241-249).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "../tnsx.h"
#include "NeighborList.h"

namespace tns
{
	class TreeNSearch
	{
	public:
		TreeNSearch() { create_(); }
		~TreeNSearch() { tnsx_destroy(ctx_); }
		// Copyable like the reference (TreeNSearch.h:36-37 declares neither copy operation): a copy is a NEW engine context with the same
		// configuration -- point sets (the same user pointers), radius / cell size, symmetric flag, active searches.  Results are not
		// carried over: run() on the copy produces them.  (The reference's implicit copy shares its executor pointer with the original.)
		TreeNSearch(const TreeNSearch& o) { create_(); replay_(o); }
		TreeNSearch& operator=(const TreeNSearch& o)
		{
			if (this != &o) { tnsx_destroy(ctx_); ctx_ = nullptr; views_.clear(); n_sets_at_run_ = 0; zsort_.clear(); zsort_fetched_.clear(); log_ = Log(); create_(); replay_(o); }
			return *this;
		}

	private:
		void create_()
		{
			tnsx_options opt;
			tnsx_default_options(&opt);
			opt.mirror_to_host = 1;
			// TNSX_DEVICES="0,1,2,3": shard every run over these GPUs (multi-device mode of the engine, include/tnsx.h) -- the
			// knob for callers whose source cannot change; one ordinal (or nothing) = one GPU
			if (const char* e = std::getenv("TNSX_DEVICES")) {
				int n = 0;
				for (const char* q = e; *q && n < 8;) {
					char* end = nullptr;
					const long v = std::strtol(q, &end, 10);
					if (end == q) break;
					opt.device_ids[n++] = (int)v;
					q = (*end == ',') ? end + 1 : end;
				}
				if (n > 1) opt.n_devices = n; else if (n == 1) opt.device_id = opt.device_ids[0];
			}
			if (tnsx_create(&opt, &ctx_) != TNSX_OK) {
				std::cout << "tns::TreeNSearch error: " << tnsx_last_error(nullptr) << std::endl;
				exit(-1);
			}
		}
		// what a copy replays (the engine keeps the same state; the shim keeps it too, in the order the calls were made)
		struct SetRec { const void* xyz; const void* radii; int n; unsigned flags_xyz, flags_radii; };
		struct Log { std::vector<SetRec> sets; bool radius_set = false; float radius = 0.f; bool cell_set = false; float cell = 0.f; bool symmetric = true; };
		int add_(const void* xyz, const void* radii, const int n, const unsigned flags)
		{
			const int id = id_(tnsx_add_point_set(ctx_, xyz, radii, n, flags));
			log_.sets.push_back({ xyz, radii, n, flags, flags });
			return id;
		}
		void resize_(const int set_id, const void* xyz, const void* radii, const int n, const unsigned flags)
		{
			ok_(tnsx_resize_point_set(ctx_, set_id, xyz, radii, n, flags));
			SetRec& r = log_.sets[(size_t)set_id];
			r.xyz = xyz; r.n = n; r.flags_xyz = flags;
			if (radii || (flags & TNSX_VARIABLE)) { r.radii = radii; r.flags_radii = flags; }
		}
		void replay_(const TreeNSearch& o)
		{
			if (o.log_.radius_set) set_search_radius(o.log_.radius);
			if (o.log_.cell_set) set_cell_size(o.log_.cell);
			for (const SetRec& r : o.log_.sets) {
				const int id = add_(r.xyz, r.radii, r.n, r.flags_radii);
				if (r.flags_xyz != r.flags_radii) resize_(id, r.xyz, nullptr, r.n, r.flags_xyz & ~TNSX_VARIABLE);   // (points of another element type than the radii)
			}
			set_symmetric_search(o.log_.symmetric);
			const int ns = o.get_n_sets();
			for (int i = 0; i < ns; i++) for (int j = 0; j < ns; j++) if (o.is_search_active(i, j)) set_active_search(i, j, true);
			n_threads_ = o.n_threads_;
		}

	public:
		// ----------------------------------------------------------------------------- main interface
		// fixed-radius sets (TreeNSearch.h:50, :63)
		int add_point_set(const float* points_begin, const int n_points) { return add_(points_begin, nullptr, n_points, TNSX_F32 | TNSX_HOST); }
		int add_point_set(const double* points_begin, const int n_points) { return add_(points_begin, nullptr, n_points, TNSX_F64 | TNSX_HOST); }
		void resize_point_set(const int set_id, const float* points_begin, const int n_points) { resize_(set_id, points_begin, nullptr, n_points, TNSX_F32 | TNSX_HOST); }
		void resize_point_set(const int set_id, const double* points_begin, const int n_points) { resize_(set_id, points_begin, nullptr, n_points, TNSX_F64 | TNSX_HOST); }
		void set_search_radius(const float search_radius) { ok_(tnsx_set_search_radius(ctx_, search_radius)); log_.radius_set = true; log_.radius = search_radius; }
		void set_search_radius(const double search_radius) { set_search_radius((float)search_radius); }

		// variable-radius sets (TreeNSearch.h:112, :126)
		int add_point_set(const float* points_begin, const float* radii_begin, const int n_points) { return add_(points_begin, radii_begin, n_points, TNSX_F32 | TNSX_HOST | TNSX_VARIABLE); }
		int add_point_set(const double* points_begin, const double* radii_begin, const int n_points) { return add_(points_begin, radii_begin, n_points, TNSX_F64 | TNSX_HOST | TNSX_VARIABLE); }
		void resize_point_set(const int set_id, const float* points_begin, const float* radii_begin, const int n_points) { resize_(set_id, points_begin, radii_begin, n_points, TNSX_F32 | TNSX_HOST | TNSX_VARIABLE); }
		void resize_point_set(const int set_id, const double* points_begin, const double* radii_begin, const int n_points) { resize_(set_id, points_begin, radii_begin, n_points, TNSX_F64 | TNSX_HOST | TNSX_VARIABLE); }

		void set_cell_size(const float cell_size) { ok_(tnsx_set_cell_size(ctx_, cell_size)); log_.cell_set = true; log_.cell = cell_size; }
		void set_cell_size(const double cell_size) { set_cell_size((float)cell_size); }

		/** Build + query on the GPU; lists are complete (and mirrored to the host) on return. */
		void run()
		{
			ok_(tnsx_run(ctx_));
			refresh_views_();
		}
		/** The reference's scalar twin (double accumulation) is not a separate code path here; its world box is
		 *  (TreeNSearch.cpp:415-522: the tight box, without the origin that run()'s SIMD remainder loop adds). */
		void run_scalar()
		{
			ok_(tnsx_run_scalar(ctx_));
			refresh_views_();
		}

		NeighborList get_neighborlist(const int set_i, const int set_j, const int point_i) const
		{
			// (the reference only asserts here, TreeNSearch.cpp:243-246, and dereferences a null pointer in release builds when the
			//  pair was not active at the last run; this shim says so and exits like every other misuse does)
			if (set_i < 0 || set_j < 0 || set_i >= n_sets_at_run_ || set_j >= n_sets_at_run_ ||
			    views_[(size_t)set_i * (size_t)n_sets_at_run_ + (size_t)set_j].records == nullptr) {
				std::cout << "tns::TreeNSearch::get_neighborlist error: no neighbour lists for (" << set_i << " -> " << set_j
				          << "): the search was not active at the last run()." << std::endl;
				exit(-1);
			}
			const View& v = views_[(size_t)set_i * (size_t)n_sets_at_run_ + (size_t)set_j];
			return NeighborList(v.records + v.offsets[point_i]);
		}
		template<typename FUNC>
		inline void for_each_neighbor(const int set_i, const int set_j, const int i, FUNC f)
		{
			const NeighborList nl = this->get_neighborlist(set_i, set_j, i);
			const int n = nl.size();
			for (int k = 0; k < n; k++) f(nl[k]);
		}

		void prepare_zsort()
		{
			ok_(tnsx_prepare_zsort(ctx_));
			// the orders stay in HBM; a set's order is copied to the host when somebody asks for it (order_of_)
			zsort_.assign((size_t)tnsx_get_n_sets(ctx_), std::vector<int>());
			zsort_fetched_.assign(zsort_.size(), 0);
		}
		template<typename T>
		void apply_zsort(const int set_i, T* data_ptr, const int stride = 1) const
		{
			if (!this->does_set_exist(set_i)) {
				std::cout << "tns::TreeNSearch::apply_zsort error: set to z_sort does not exit." << std::endl;
				exit(-1);
			}
			if ((size_t)set_i >= zsort_.size()) {
				std::cout << "tns::TreeNSearch::apply_zsort error: no zsort order ready for set_i (" << set_i << ")." << std::endl;
				exit(-1);
			}
			const std::vector<int>& map = order_of_(set_i);
			const size_t n = map.size();
			const size_t st = (size_t)stride;
			std::vector<T> old(data_ptr, data_ptr + n * st);
			#pragma omp parallel for schedule(static)
			for (long long k = 0; k < (long long)n; k++) {
				const size_t src = (size_t)map[(size_t)k] * st;
				for (size_t j = 0; j < st; j++) data_ptr[(size_t)k * st + j] = old[src + j];
			}
		}
		void set_symmetric_search(const bool activate) { ok_(tnsx_set_symmetric_search(ctx_, activate ? 1 : 0)); log_.symmetric = activate; }

		// ----------------------------------------------------------------------------- secondary methods
		void print_state() const
		{
			tnsx_stats s;
			tnsx_get_stats(ctx_, &s);
			std::cout << "tnsx state: sets " << s.n_sets << ", points " << s.n_points << ", queries " << s.n_queries << ", neighbours " << s.n_neighbors
			          << ", search grid " << s.grid_dims[0] << "x" << s.grid_dims[1] << "x" << s.grid_dims[2] << " (cell " << s.grid_cell_size
			          << "), occupied cells " << s.n_occupied_cells << ", world cells/dim " << s.world_cells_pow2 << std::endl;
		}
		uint64_t get_neighborlist_n_bytes() const { return tnsx_get_neighborlist_n_bytes(ctx_); }

		// ----------------------------------------------------------------------------- setters and getters
		void set_all_searches(const bool active) { ok_(tnsx_set_all_searches(ctx_, active ? 1 : 0)); }
		void set_active_search(const int set_i, const int set_j, const bool active = true) { ok_(tnsx_set_active_search(ctx_, set_i, set_j, active ? 1 : 0)); }
		void set_active_search(const int set_i, const bool search_in_all = true, const bool be_found_by_all = true) { ok_(tnsx_set_active_search_all(ctx_, set_i, search_in_all ? 1 : 0, be_found_by_all ? 1 : 0)); }
		// CPU tuning knobs of the reference: accepted, meaningless on the GPU
		void set_n_threads(const int n_threads) { n_threads_ = n_threads; }
		void set_recursion_cap(const int) {}
		void set_n_points_for_parallel_octree(const int = 200000) {}

		int get_n_sets() const { return tnsx_get_n_sets(ctx_); }
		int get_n_threads() const { return n_threads_; }
		int get_n_points_in_set(const int set_i) const { return tnsx_get_n_points_in_set(ctx_, set_i); }
		int get_total_n_points() const { return (int)tnsx_get_total_n_points(ctx_); }
		bool is_search_active(const int set_i, const int set_j) const { return tnsx_is_search_active(ctx_, set_i, set_j) != 0; }
		bool does_set_exist(const int set_i) const { return tnsx_does_set_exist(ctx_, set_i) != 0; }
		const std::vector<int>& get_zsort_order(const int set_i) const { return order_of_(set_i); }

		/** Extension: the underlying engine handle (device-side CSR views, stats, arithmetic mode). */
		tnsx_context* engine() const { return ctx_; }

	private:
		struct View { const uint64_t* offsets = nullptr; const int* records = nullptr; };

		void ok_(const tnsx_status st) const
		{
			if (st != TNSX_OK) {
				std::cout << tnsx_last_error(ctx_) << std::endl;
				exit(-1);
			}
		}
		int id_(const int id_or_neg_status) const
		{
			if (id_or_neg_status < 0) ok_((tnsx_status)(-id_or_neg_status));
			return id_or_neg_status;
		}
		// host copy of one set's z-order, fetched from the engine on first use (the engine itself keeps it on the device)
		const std::vector<int>& order_of_(const int set_i) const
		{
			if (set_i < 0 || (size_t)set_i >= zsort_.size()) {   // (get_zsort_order before prepare_zsort: the reference indexes an empty vector)
				std::cout << "tns::TreeNSearch::apply_zsort error: no zsort order ready for set_i (" << set_i << ")." << std::endl;
				exit(-1);
			}
			if (!zsort_fetched_[(size_t)set_i]) {
				#pragma omp critical(tnsx_zsort_fetch)
				if (!zsort_fetched_[(size_t)set_i]) {
					const int* host = nullptr; int n = 0;
					ok_(tnsx_get_zsort_order(ctx_, set_i, &host, nullptr, &n));
					zsort_[(size_t)set_i].assign(host, host + n);
					zsort_fetched_[(size_t)set_i] = 1;
				}
			}
			return zsort_[(size_t)set_i];
		}
		void refresh_views_()
		{
			n_sets_at_run_ = tnsx_get_n_sets(ctx_);
			views_.assign((size_t)n_sets_at_run_ * (size_t)n_sets_at_run_, View());
			for (int i = 0; i < n_sets_at_run_; i++) {
				for (int j = 0; j < n_sets_at_run_; j++) {
					if (!tnsx_is_search_active(ctx_, i, j)) continue;
					tnsx_csr_view v;
					ok_(tnsx_get_pair_view(ctx_, i, j, &v));
					views_[(size_t)i * (size_t)n_sets_at_run_ + (size_t)j].offsets = v.offsets_host;
					views_[(size_t)i * (size_t)n_sets_at_run_ + (size_t)j].records = v.records_host;
				}
			}
		}

		tnsx_context* ctx_ = nullptr;
		std::vector<View> views_;
		int n_sets_at_run_ = 0;
		mutable std::vector<std::vector<int>> zsort_;
		mutable std::vector<char> zsort_fetched_;
		int n_threads_ = -1;
		Log log_;
	};
}
