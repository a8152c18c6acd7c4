/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of the reference's fixed-radius
 * neighbour search path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product path (treensearch_amd/, include/) never does.
 *
 * Parity pinning: tests/golden/make_golden.py runs the REAL reference built from /root/reference
 * (oracle/_ref/libtns_ref*.so, see oracle/Makefile) on every seeded case, asserts that this
 * restatement agrees with it list by list, and writes the fixtures in tests/golden/;
 * tests/test_oracle_golden.py checks the restatement against those fixtures wherever the tests run.
 *
 * What is restated (reference file:line, relative to /root/reference):
 *   - the neighbour predicate of the AVX2 path and of tests/BruteforceNSearch:
 *       TreeNSearch/source/TreeNSearch.cpp:2478-2486 (asymmetric), :2538-2547 (symmetric),
 *       tests/BruteforceNSearch.cpp:82-100; r2 = r*r in fp32 (TreeNSearch.cpp:29, :2352; BF.cpp:82,91)
 *   - self exclusion by index (TreeNSearch.cpp:2465-2466; BF.cpp:86), set-local indices (:2260)
 *   - world box update (TreeNSearch.cpp:415-522 == :523-645 after the reduction)
 *   - cell quantisation (TreeNSearch.cpp:713-715) and Morton interleave x->bit0,y->bit1,z->bit2
 *     (extern/libmorton, used at TreeNSearch.cpp:2617, :2693)
 * What is NOT restated: the octree.  Candidate generation here is a plain uniform grid (or all
 * pairs); only the *sets* must agree, and they are data-structure independent.
 *
 * Two arithmetic modes:
 *   STRICT      d2 = ((dx*dx + dy*dy) + dz*dz)            every op rounded (source-literal; what the
 *                                                          reference gives with -ffp-contract=off)
 *   CONTRACTED  d2 = fmaf(dz,dz, fmaf(dx,dx, dy*dy))      what GCC 11.4 emits for both reference files
 *                                                          under the reference's own flags
 * This file MUST be compiled with -ffp-contract=off (oracle/Makefile does) so that STRICT is strict.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TNSO_STRICT 0
#define TNSO_CONTRACTED 1

typedef struct tnso_result {
	int n;              /* number of query points */
	int64_t total;      /* total number of neighbour indices */
	int64_t* offsets;   /* n+1, exclusive scan of counts */
	int* indices;       /* total, each list ascending in j */
} tnso_result;

/* ------------------------------------------------------------------------------------------ */
/* predicate                                                                                  */
/* ------------------------------------------------------------------------------------------ */
static inline float dist_sq(const float* p, const float* q, int mode)
{
	const float dx = p[0] - q[0];
	const float dy = p[1] - q[1];
	const float dz = p[2] - q[2];
	if (mode == TNSO_STRICT) {
		const float a = dx * dx;
		const float b = dy * dy;
		const float s = a + b;
		const float c = dz * dz;
		return s + c;
	}
	else {
		const float t = dy * dy;
		return fmaf(dz, dz, fmaf(dx, dx, t));
	}
}

/* exported for unit tests of the arithmetic */
float tnso_dist_sq(const float* p, const float* q, int mode) { return dist_sq(p, q, mode); }

/* ------------------------------------------------------------------------------------------ */
/* small growable int vector                                                                  */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int* d; int64_t n, cap; } ivec;
static void ivec_push(ivec* v, int x)
{
	if (v->n == v->cap) {
		v->cap = v->cap ? v->cap * 2 : 1024;
		v->d = (int*)realloc(v->d, sizeof(int) * (size_t)v->cap);
	}
	v->d[v->n++] = x;
}
static int cmp_int(const void* a, const void* b)
{
	const int x = *(const int*)a, y = *(const int*)b;
	return (x > y) - (x < y);
}

/* ------------------------------------------------------------------------------------------ */
/* pair search: queries = set A, candidates = set B                                           */
/*   ra/rb == NULL  -> fixed radius mode (r2 = fixed_radius*fixed_radius, no symmetric check,  */
/*                     TreeNSearch.cpp:2431: symmetric only if radii are per point)            */
/*   same_set       -> a point never neighbours itself (index equality)                        */
/*   use_grid == 0  -> all pairs (restates BruteforceNSearch::run)                             */
/* ------------------------------------------------------------------------------------------ */
tnso_result* tnso_pair_search(const float* xa, const float* ra, int na,
                              const float* xb, const float* rb, int nb,
                              float fixed_radius, int symmetric, int same_set, int mode, int use_grid)
{
	tnso_result* res = (tnso_result*)calloc(1, sizeof(tnso_result));
	res->n = na;
	res->offsets = (int64_t*)calloc((size_t)na + 1, sizeof(int64_t));
	const int variable = (ra != NULL);
	const int sym = variable && symmetric;
	const float r2_fixed = fixed_radius * fixed_radius;

	/* ---- candidate grid over B (double arithmetic + 0.1 % margin: completeness is independent of
	        fp32 rounding of any binning; the predicate alone decides membership) ---- */
	double rmax = 0.0;
	if (variable) {
		for (int i = 0; i < na; i++) if (ra[i] > rmax) rmax = ra[i];
		if (sym) for (int j = 0; j < nb; j++) if (rb[j] > rmax) rmax = rb[j];
	}
	else {
		rmax = fixed_radius;
	}
	double lo[3] = { DBL_MAX, DBL_MAX, DBL_MAX }, hi[3] = { -DBL_MAX, -DBL_MAX, -DBL_MAX };
	for (int j = 0; j < nb; j++) for (int d = 0; d < 3; d++) {
		const double v = xb[3 * (size_t)j + d];
		if (v < lo[d]) lo[d] = v;
		if (v > hi[d]) hi[d] = v;
	}
	int* cell_start = NULL; int* cell_pts = NULL;
	int64_t dims[3] = { 1, 1, 1 };
	double h = rmax * 1.001;
	if (use_grid && nb > 0 && h > 0.0) {
		/* coarsen until the dense table is affordable */
		for (;;) {
			int64_t tot = 1;
			for (int d = 0; d < 3; d++) { dims[d] = (int64_t)floor((hi[d] - lo[d]) / h) + 1; tot *= dims[d]; }
			if (tot <= (int64_t)1 << 27) break;
			h *= 1.5;
		}
		const int64_t ncell = dims[0] * dims[1] * dims[2];
		cell_start = (int*)calloc((size_t)ncell + 1, sizeof(int));
		cell_pts = (int*)malloc(sizeof(int) * (size_t)nb);
		int64_t* cid = (int64_t*)malloc(sizeof(int64_t) * (size_t)nb);
		for (int j = 0; j < nb; j++) {
			int64_t c[3];
			for (int d = 0; d < 3; d++) {
				c[d] = (int64_t)floor((xb[3 * (size_t)j + d] - lo[d]) / h);
				if (c[d] < 0) c[d] = 0;
				if (c[d] >= dims[d]) c[d] = dims[d] - 1;
			}
			cid[j] = (c[2] * dims[1] + c[1]) * dims[0] + c[0];
			cell_start[cid[j] + 1]++;
		}
		for (int64_t c = 0; c < ncell; c++) cell_start[c + 1] += cell_start[c];
		int* cur = (int*)malloc(sizeof(int) * (size_t)ncell);
		memcpy(cur, cell_start, sizeof(int) * (size_t)ncell);
		for (int j = 0; j < nb; j++) cell_pts[cur[cid[j]]++] = j;   /* ascending j inside a cell */
		free(cur); free(cid);
	}
	else {
		use_grid = 0;
	}

	int nthreads = 1;
#ifdef _OPENMP
	nthreads = omp_get_max_threads();
#endif
	ivec* tl = (ivec*)calloc((size_t)nthreads, sizeof(ivec));
	int* chunk_begin = (int*)calloc((size_t)nthreads + 1, sizeof(int));
	int* counts = (int*)calloc((size_t)na + 1, sizeof(int));

	#pragma omp parallel num_threads(nthreads)
	{
		int tid = 0, nt = 1;
#ifdef _OPENMP
		tid = omp_get_thread_num(); nt = omp_get_num_threads();
#endif
		const int begin = (int)((int64_t)na * tid / nt);
		const int end = (int)((int64_t)na * (tid + 1) / nt);
		chunk_begin[tid] = begin;
		ivec* out = &tl[tid];
		for (int i = begin; i < end; i++) {
			const float* p = xa + 3 * (size_t)i;
			const float r2i = variable ? ra[i] * ra[i] : r2_fixed;
			const int64_t list_begin = out->n;
			if (!use_grid) {
				for (int j = 0; j < nb; j++) {
					if (same_set && i == j) continue;
					const float d2 = dist_sq(p, xb + 3 * (size_t)j, mode);
					int hit = d2 <= r2i;
					if (sym) { const float r2j = rb[j] * rb[j]; hit = hit || (d2 <= r2j); }
					if (hit) ivec_push(out, j);
				}
			}
			else {
				int64_t c[3];
				for (int d = 0; d < 3; d++) c[d] = (int64_t)floor((p[d] - lo[d]) / h);
				for (int64_t cz = c[2] - 1; cz <= c[2] + 1; cz++) {
					if (cz < 0 || cz >= dims[2]) continue;
					for (int64_t cy = c[1] - 1; cy <= c[1] + 1; cy++) {
						if (cy < 0 || cy >= dims[1]) continue;
						for (int64_t cx = c[0] - 1; cx <= c[0] + 1; cx++) {
							if (cx < 0 || cx >= dims[0]) continue;
							const int64_t cc = (cz * dims[1] + cy) * dims[0] + cx;
							for (int k = cell_start[cc]; k < cell_start[cc + 1]; k++) {
								const int j = cell_pts[k];
								if (same_set && i == j) continue;
								const float d2 = dist_sq(p, xb + 3 * (size_t)j, mode);
								int hit = d2 <= r2i;
								if (sym) { const float r2j = rb[j] * rb[j]; hit = hit || (d2 <= r2j); }
								if (hit) ivec_push(out, j);
							}
						}
					}
				}
				qsort(out->d + list_begin, (size_t)(out->n - list_begin), sizeof(int), cmp_int);
			}
			counts[i] = (int)(out->n - list_begin);
		}
	}
	for (int i = 0; i < na; i++) res->offsets[i + 1] = res->offsets[i] + counts[i];
	res->total = res->offsets[na];
	res->indices = (int*)malloc(sizeof(int) * (size_t)(res->total > 0 ? res->total : 1));
	for (int t = 0; t < nthreads; t++) {
		if (tl[t].n) memcpy(res->indices + res->offsets[chunk_begin[t]], tl[t].d, sizeof(int) * (size_t)tl[t].n);
		free(tl[t].d);
	}
	free(tl); free(chunk_begin); free(counts); free(cell_start); free(cell_pts);
	return res;
}

void tnso_result_free(tnso_result* r)
{
	if (!r) return;
	free(r->offsets); free(r->indices); free(r);
}

/* ------------------------------------------------------------------------------------------ */
/* order-independent digest of a CSR neighbour structure                                      */
/*   per point: FNV-1a-64 over the u32 words (p, count, sorted j...) ; out[0]=sum, out[1]=xor  */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t fnv_word(uint64_t h, uint32_t w)
{
	for (int b = 0; b < 4; b++) { h ^= (w >> (8 * b)) & 0xffu; h *= 0x100000001b3ull; }
	return h;
}
void tnso_digest_csr(int n, const int64_t* offsets, const int* indices, int already_sorted, uint64_t* out)
{
	uint64_t sum = 0, x = 0;
	#pragma omp parallel
	{
		uint64_t lsum = 0, lx = 0;
		int* tmp = NULL; int64_t cap = 0;
		#pragma omp for schedule(static)
		for (int p = 0; p < n; p++) {
			const int64_t b = offsets[p], cnt = offsets[p + 1] - b;
			const int* l = indices + b;
			if (!already_sorted) {
				if (cnt > cap) { cap = cnt * 2; tmp = (int*)realloc(tmp, sizeof(int) * (size_t)cap); }
				memcpy(tmp, l, sizeof(int) * (size_t)cnt);
				qsort(tmp, (size_t)cnt, sizeof(int), cmp_int);
				l = tmp;
			}
			uint64_t hsh = 0xcbf29ce484222325ull;
			hsh = fnv_word(hsh, (uint32_t)p);
			hsh = fnv_word(hsh, (uint32_t)cnt);
			for (int64_t k = 0; k < cnt; k++) hsh = fnv_word(hsh, (uint32_t)l[k]);
			lsum += hsh; lx ^= hsh;
		}
		free(tmp);
		#pragma omp critical
		{ sum += lsum; x ^= lx; }
	}
	out[0] = sum; out[1] = x;
}

/* ------------------------------------------------------------------------------------------ */
/* world box (TreeNSearch.cpp:415-522).  box = {bottom[3], top[3]} persistent across calls;   */
/* initial state bottom=+FLT_MAX, top=-FLT_MAX (octree_internals.h:29-30).                     */
/* tight = {min[3], max[3]} of all points (+FLT_MAX / -FLT_MAX when there are no points).      */
/* returns 0 = box kept, 1 = box recomputed, -1 = more than 32768 cells per dimension          */
/* ------------------------------------------------------------------------------------------ */
int tnso_world_box_update(float* box, const float* tight, float cell_size, int* n_cells_pow2_out)
{
	float* bottom = box; float* top = box + 3;
	const float* nb = tight; const float* nt = tight + 3;
	if (bottom[0] <= nb[0] && nt[0] <= top[0] &&
	    bottom[1] <= nb[1] && nt[1] <= top[1] &&
	    bottom[2] <= nb[2] && nt[2] <= top[2]) {
		return 0;
	}
	for (int d = 0; d < 3; d++) { bottom[d] = nb[d]; top[d] = nt[d]; }
	float center[3];
	for (int d = 0; d < 3; d++) center[d] = 0.5f * (top[d] + bottom[d]);
	float length = 0.0f;
	for (int d = 0; d < 3; d++) { const float e = top[d] - bottom[d]; if (e > length) length = e; }
	length += 100.0f * FLT_EPSILON;
	length *= 1.1f;
	const int n_cells = (int)(length / cell_size) + 1;
	int n_pow2 = 1;
	while (n_pow2 < n_cells) n_pow2 *= 2;
	length = cell_size * (float)n_pow2;
	if (n_cells_pow2_out) *n_cells_pow2_out = n_pow2;
	if (n_pow2 > 32768) return -1;
	for (int d = 0; d < 3; d++) {
		bottom[d] = center[d] - 0.5f * length;
		top[d] = center[d] + 0.5f * length;
	}
	return 1;
}

/* tight bounds of a point array, merged into tight[6] (caller initialises to +/-FLT_MAX) */
void tnso_tight_bounds(const float* x, int n, float* tight)
{
	for (int i = 0; i < n; i++) for (int d = 0; d < 3; d++) {
		const float v = x[3 * (size_t)i + d];
		if (v < tight[d]) tight[d] = v;
		if (v > tight[3 + d]) tight[3 + d] = v;
	}
}

/* Tight bounds as run() -- the AVX2 path -- sees them (TreeNSearch.cpp:523-592, _update_world_AABB_simd).  Every thread takes
 * its points two at a time as [x y z x y z . .] (:558-562) and the LAST 2..3 points of its chunk one at a time as
 * [x y z 0 0 0 0 0] (:564-569, _mm256_setr_ps with zeros): the zeros sit in lanes 3..5, which the final reduction folds into
 * the result (:587-590: min(b[d], b[3 + d]), max(t[d], t[3 + d])).  A thread with at least one point always takes the
 * remainder loop (the pair loop stops at end - 3), and at least one thread has points whenever the set has any.  So the
 * "tight" box of run() and of the no-tree path of prepare_zsort() (:2674) is the tight box of the points UNITED WITH THE
 * ORIGIN; run_scalar() (:415-472) has no such padding and uses tnso_tight_bounds.  Pinned against the reference itself by
 * tests/golden (world blocks) and tests/test_oracle_golden.py. */
void tnso_tight_bounds_simd(const float* x, int n, float* tight)
{
	tnso_tight_bounds(x, n, tight);
	if (n > 0) for (int d = 0; d < 3; d++) {
		if (0.0f < tight[d]) tight[d] = 0.0f;
		if (0.0f > tight[3 + d]) tight[3 + d] = 0.0f;
	}
}

/* ------------------------------------------------------------------------------------------ */
/* z-sort                                                                                     */
/*   cell coords: (uint)((p - bottom) * cell_size_inv)  fp32 sub, mul, truncate (TNS.cpp:713)  */
/*   key: Morton interleave, x -> bit 0, y -> bit 1, z -> bit 2 (libmorton)                    */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t spread3(uint64_t v)
{
	v &= 0x1fffffull;
	v = (v | (v << 32)) & 0x1f00000000ffffull;
	v = (v | (v << 16)) & 0x1f0000ff0000ffull;
	v = (v | (v << 8)) & 0x100f00f00f00f00full;
	v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
	v = (v | (v << 2)) & 0x1249249249249249ull;
	return v;
}
uint64_t tnso_morton3(uint32_t x, uint32_t y, uint32_t z) { return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2); }

void tnso_zsort_keys(const float* x, int n, const float* bottom, float cell_size_inv, uint64_t* keys)
{
	#pragma omp parallel for schedule(static)
	for (int i = 0; i < n; i++) {
		const float fx = (x[3 * (size_t)i + 0] - bottom[0]) * cell_size_inv;
		const float fy = (x[3 * (size_t)i + 1] - bottom[1]) * cell_size_inv;
		const float fz = (x[3 * (size_t)i + 2] - bottom[2]) * cell_size_inv;
		keys[i] = tnso_morton3((uint32_t)fx, (uint32_t)fy, (uint32_t)fz);
	}
}

/* the engine's z-sort definition: stable sort by the cell-level Morton key -> new_to_old */
typedef struct { uint64_t k; int i; } kv;
static int cmp_kv(const void* a, const void* b)
{
	const kv* x = (const kv*)a; const kv* y = (const kv*)b;
	if (x->k != y->k) return (x->k > y->k) - (x->k < y->k);
	return (x->i > y->i) - (x->i < y->i);
}
void tnso_zsort_order(const float* x, int n, const float* bottom, float cell_size_inv, int* new_to_old)
{
	uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
	kv* a = (kv*)malloc(sizeof(kv) * (size_t)(n > 0 ? n : 1));
	tnso_zsort_keys(x, n, bottom, cell_size_inv, keys);
	for (int i = 0; i < n; i++) { a[i].k = keys[i]; a[i].i = i; }
	qsort(a, (size_t)n, sizeof(kv), cmp_kv);
	for (int i = 0; i < n; i++) new_to_old[i] = a[i].i;
	free(keys); free(a);
}

/* checks that new_to_old is a permutation of 0..n-1 and that keys[new_to_old[.]] is non-decreasing.
   returns 0 ok, 1 not a permutation, 2 not Morton ordered */
int tnso_check_zsort(const uint64_t* keys, const int* new_to_old, int n)
{
	unsigned char* seen = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
	int rc = 0;
	for (int i = 0; i < n && !rc; i++) {
		const int o = new_to_old[i];
		if (o < 0 || o >= n || seen[o]) rc = 1; else seen[o] = 1;
	}
	for (int i = 1; i < n && !rc; i++) if (keys[new_to_old[i - 1]] > keys[new_to_old[i]]) rc = 2;
	free(seen);
	return rc;
}


/* ------------------------------------------------------------------------------------------ */
/* CSR of a permuted copy -> CSR in the original index space.                                 */
/*   new_to_old[k] = original index of the point at position k of the permuted copy.          */
/*   in: offsets_p/indices_p over permuted positions; out: offsets_o (n+1) / indices_o with    */
/*   every list ascending.                                                                     */
/* ------------------------------------------------------------------------------------------ */
void tnso_remap_csr(int n, const int* new_to_old_i, const int* new_to_old_j,
                    const int64_t* offsets_p, const int* indices_p, int64_t* offsets_o, int* indices_o)
{
	int* old_to_new = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
	for (int k = 0; k < n; k++) old_to_new[new_to_old_i[k]] = k;
	offsets_o[0] = 0;
	for (int o = 0; o < n; o++) {
		const int k = old_to_new[o];
		offsets_o[o + 1] = offsets_o[o] + (offsets_p[k + 1] - offsets_p[k]);
	}
	#pragma omp parallel for schedule(static)
	for (int o = 0; o < n; o++) {
		const int k = old_to_new[o];
		const int64_t cnt = offsets_p[k + 1] - offsets_p[k];
		int* dst = indices_o + offsets_o[o];
		const int* src = indices_p + offsets_p[k];
		for (int64_t t = 0; t < cnt; t++) dst[t] = new_to_old_j[src[t]];
		qsort(dst, (size_t)cnt, sizeof(int), cmp_int);
	}
	free(old_to_new);
}

int tnso_num_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}
