// TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
//
// Flat C wrapper around the *real* reference (tns::TreeNSearch and tests/BruteforceNSearch)
// so that Python tests / the golden generator / bench.py's cpu_baseline leg can drive the
// reference through ctypes.  This file contains no reference source: it includes the reference
// headers where they lie under $(REF)=/root/reference and is compiled together with
//   $(REF)/TreeNSearch/source/TreeNSearch.cpp   and   $(REF)/tests/BruteforceNSearch.cpp
// by oracle/Makefile into oracle/_ref/libtns_ref*.so (git-ignored, travels to the GPU box).
//
// Interfaces wrapped: TreeNSearch.h:28-427 (public API), BruteforceNSearch.h:17-51.
// The world box (domain_float, TreeNSearch.h:400) and the cell size (:398) are private members with no getter; the accessors
// at the end of this file read them.  Every standard header the reference pulls in is included FIRST, so that the keyword
// override below touches the reference's own class declarations only (same layout, same code: an access specifier changes
// neither), in this one translation unit.
#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <limits>
#include <numeric>
#include <string>
#include <vector>
#include <immintrin.h>
#include <malloc.h>
#include <omp.h>
#define private public
#include <TreeNSearch>
#undef private
#include "BruteforceNSearch.h"

extern "C" {

// ----------------------------------------------------------------------------- tns::TreeNSearch
void* ref_tns_create() { return new tns::TreeNSearch(); }
void ref_tns_destroy(void* h) { delete static_cast<tns::TreeNSearch*>(h); }

#define TNS(h) (static_cast<tns::TreeNSearch*>(h))

int ref_tns_add_point_set_f(void* h, const float* xyz, const float* radii, int n)
{
	return radii ? TNS(h)->add_point_set(xyz, radii, n) : TNS(h)->add_point_set(xyz, n);
}
int ref_tns_add_point_set_d(void* h, const double* xyz, const double* radii, int n)
{
	return radii ? TNS(h)->add_point_set(xyz, radii, n) : TNS(h)->add_point_set(xyz, n);
}
void ref_tns_resize_point_set_f(void* h, int set, const float* xyz, const float* radii, int n)
{
	if (radii) TNS(h)->resize_point_set(set, xyz, radii, n); else TNS(h)->resize_point_set(set, xyz, n);
}
void ref_tns_resize_point_set_d(void* h, int set, const double* xyz, const double* radii, int n)
{
	if (radii) TNS(h)->resize_point_set(set, xyz, radii, n); else TNS(h)->resize_point_set(set, xyz, n);
}
void ref_tns_set_search_radius(void* h, float r) { TNS(h)->set_search_radius(r); }
void ref_tns_set_cell_size(void* h, float c) { TNS(h)->set_cell_size(c); }
void ref_tns_set_symmetric_search(void* h, int on) { TNS(h)->set_symmetric_search(on != 0); }
void ref_tns_set_active_search(void* h, int i, int j, int on) { TNS(h)->set_active_search(i, j, on != 0); }
void ref_tns_set_active_search_all(void* h, int i, int search, int found) { TNS(h)->set_active_search(i, search != 0, found != 0); }
void ref_tns_set_all_searches(void* h, int on) { TNS(h)->set_all_searches(on != 0); }
void ref_tns_set_n_threads(void* h, int n) { TNS(h)->set_n_threads(n); }
void ref_tns_set_recursion_cap(void* h, int n) { TNS(h)->set_recursion_cap(n); }
void ref_tns_run(void* h) { TNS(h)->run(); }
void ref_tns_run_scalar(void* h) { TNS(h)->run_scalar(); }
void ref_tns_prepare_zsort(void* h) { TNS(h)->prepare_zsort(); }
int ref_tns_get_n_sets(void* h) { return TNS(h)->get_n_sets(); }
int ref_tns_get_n_points_in_set(void* h, int s) { return TNS(h)->get_n_points_in_set(s); }
int ref_tns_is_search_active(void* h, int i, int j) { return TNS(h)->is_search_active(i, j) ? 1 : 0; }

// copy of the zsort new->old map of one set
void ref_tns_get_zsort_order(void* h, int set, int* out)
{
	const std::vector<int>& v = TNS(h)->get_zsort_order(set);
	std::memcpy(out, v.data(), sizeof(int) * v.size());
}
// apply_zsort<T> for T = float / int / double (the instantiations the tests need)
void ref_tns_apply_zsort_f(void* h, int set, float* data, int stride) { TNS(h)->apply_zsort(set, data, stride); }
void ref_tns_apply_zsort_i(void* h, int set, int* data, int stride) { TNS(h)->apply_zsort(set, data, stride); }
void ref_tns_apply_zsort_d(void* h, int set, double* data, int stride) { TNS(h)->apply_zsort(set, data, stride); }

// neighbour counts of pair (i,j): counts[p] = get_neighborlist(i,j,p).size(); returns total
int64_t ref_tns_get_counts(void* h, int i, int j, int* counts)
{
	const int n = TNS(h)->get_n_points_in_set(i);
	int64_t total = 0;
	#pragma omp parallel for schedule(static) reduction(+:total)
	for (int p = 0; p < n; p++) {
		counts[p] = TNS(h)->get_neighborlist(i, j, p).size();
		total += counts[p];
	}
	return total;
}
// neighbour lists of pair (i,j) in CSR form; offsets has n+1 entries (exclusive scan of counts, given by caller)
void ref_tns_get_lists(void* h, int i, int j, const int64_t* offsets, int* indices, int sort_each)
{
	const int n = TNS(h)->get_n_points_in_set(i);
	#pragma omp parallel for schedule(static)
	for (int p = 0; p < n; p++) {
		const tns::NeighborList nl = TNS(h)->get_neighborlist(i, j, p);
		int* dst = indices + offsets[p];
		std::memcpy(dst, nl.get_ptr(), sizeof(int) * (size_t)nl.size());
		if (sort_each) std::sort(dst, dst + nl.size());
	}
}

// private state (see the note at the includes): world box {bottom[3], top[3]} (TreeNSearch.h:400, updated by
// _update_world_AABB[_simd], TreeNSearch.cpp:415-645) and the grid cell size (TreeNSearch.h:398)
void ref_tns_get_world_box(void* h, float* out6)
{
	for (int d = 0; d < 3; d++) { out6[d] = TNS(h)->domain_float.bottom[d]; out6[3 + d] = TNS(h)->domain_float.top[d]; }
}
float ref_tns_get_cell_size(void* h) { return TNS(h)->cell_size; }

// ----------------------------------------------------------------------------- BruteforceNSearch
void* ref_bf_create() { return new BruteforceNSearch(); }
void ref_bf_destroy(void* h) { delete static_cast<BruteforceNSearch*>(h); }
#define BF(h) (static_cast<BruteforceNSearch*>(h))

int ref_bf_add_point_set(void* h, const float* xyz, const float* radii, int n) { return BF(h)->add_point_set(xyz, radii, n); }
int ref_bf_add_point_set_r(void* h, const float* xyz, float radius, int n) { return BF(h)->add_point_set(xyz, radius, n); }
void ref_bf_resize_point_set(void* h, int s, const float* xyz, const float* radii, int n) { BF(h)->resize_point_set(s, xyz, radii, n); }
void ref_bf_resize_point_set_r(void* h, int s, const float* xyz, float radius, int n) { BF(h)->resize_point_set(s, xyz, radius, n); }
void ref_bf_set_active_search(void* h, int i, int j, int on) { BF(h)->set_active_search(i, j, on != 0); }
void ref_bf_set_all_searches(void* h, int on) { BF(h)->set_all_searches(on != 0); }
void ref_bf_set_symmetric_search(void* h, int on) { BF(h)->set_symmetric_search(on != 0); }
void ref_bf_set_n_threads(void* h, int n) { BF(h)->set_n_threads(n); }
void ref_bf_run(void* h) { BF(h)->run(); }
int64_t ref_bf_get_counts(void* h, int i, int j, int* counts)
{
	const int n = BF(h)->n_points_per_set[i];
	int64_t total = 0;
	for (int p = 0; p < n; p++) { counts[p] = BF(h)->get_n_neighbors(i, j, p); total += counts[p]; }
	return total;
}
void ref_bf_get_lists(void* h, int i, int j, const int64_t* offsets, int* indices)
{
	const int n = BF(h)->n_points_per_set[i];
	const auto& sol = BF(h)->solution[i * BF(h)->n_sets + j];
	for (int p = 0; p < n; p++) {
		std::memcpy(indices + offsets[p], sol[p].data(), sizeof(int) * sol[p].size());
	}
}
// the reference's own comparison (BruteforceNSearch.cpp:117-178); crash flag off => compares sizes only
int ref_bf_compare(void* bf, void* tns_h) { return BF(bf)->compare(*TNS(tns_h), false) ? 1 : 0; }

}  // extern "C"
