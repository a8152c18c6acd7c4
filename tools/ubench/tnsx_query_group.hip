// gfx950 kernels: THE QUERY, group formulation (round 3) -- fixed radius, pool pass.
//
// The cell kernels of tnsx_query.hip put the CANDIDATES of a cell's 27 neighbours into the lanes and broadcast the cell's query
// points one at a time: 5 + 3 vector instructions per (query, 64 candidates), 6.4 such chunks per query at 59 neighbours found --
// the instruction floor DESIGN.md section 6 measures (vector pipe 87 % busy).  This formulation turns the problem round:
//
//   * a wave takes an occupied cell and stages the candidates of its 27 neighbours (9 rows of 3 cells: 9 contiguous pieces of the
//     sorted candidate array, ~386 points at 14 points per cell) ONCE, as rows of a matrix in LDS; the cell's query points are
//     worked off in batches of GQ = 16, one query per lane % 16.
//   * the tests themselves run on the MATRIX pipe: with coordinates local to the cell (origin = its centre),
//     |c - q|^2 - r^2 <= 0  <=>  |c|^2 - 2 q.c <= r^2 - |q|^2, and the left side for 16 candidates x 16 queries is ONE
//     v_mfma_f32_16x16x4_f32 of A = rows (-2cx, -2cy, -2cz, |c|^2) and B = columns (qx, qy, qz, 1).  The result layout puts query
//     j = lane % 16 in the lane and candidate 4 * (lane / 16) + v in register v: four lanes share a query, each sees a quarter of
//     the candidates.  What is left for the vector pipe per 64 tests: one compare against the lane's threshold, the append of the
//     hits (exec <- mask, ds_write_b16 of the candidate's slot to the lane's own list, bump of the list pointer) and one v_max:
//     ~17 issue cycles instead of the 38 of the cell kernels' test-and-compact.  Lanes of a batch without a query (a cell of 14
//     points fills 14 of 16) are the price: ~550 lane-tests per query instead of 411.
//   * the MFMA value is NOT the reference's arithmetic, so it only decides what it can decide: with delta = 2^-14 h^2 (h = cell
//     edge; bound derived below) a value below T - delta is a neighbour and one above T + delta is not whatever the rounding; a
//     lane that appended a candidate inside the band (running max of the appended values, one v_max per 64 tests) re-tests the
//     entries of its list that are near the threshold with the reference's own operations on the original coordinates.
//     ~1 % of the queries at 59 neighbours.
//   * output: the four lists of a query are concatenated behind its count word in the wave's staging area (slot -> index through
//     the cell's id table), the block of the batch's 16 records leaves with coalesced stores, one pool allocation per batch.
//   * anything that does not fit -- more than G_CMAX candidates or 64 query points, a list that overflows, a point binned far outside
//     its cell (trimmed grids) -- goes to a worklist of cells that the three cell tiers of tnsx_query.hip walk afterwards; the
//     engine runs the cell kernels alone for a pair that keeps passing on more than a quarter of its cells (tnsx_engine.cpp).
//
// Error bound behind delta (u = 2^-24; the guards in the kernel enforce |lc|^2 <= 7 h^2 for candidates and |lq|^2 <= 0.8 h^2 for queries --
// 1.5 h and 0.5 h per axis; the figures below are for the looser 11 h^2 and 3 h^2):
//   local coordinates  lc = fl(c - o): relative error u each, so |lc - lq - (c - q)| <= u (|lc| + |lq|) <= 5.1 u h and
//                      | |lc - lq|^2 - |c - q|^2 | <= 2 * 5.05 h * 5.1 u h = 52 u h^2
//   |lc|^2, |lq|^2     three products, two sums each: 3 u * 11 h^2 + 3 u * 3 h^2 = 42 u h^2
//   the MFMA           four products summed, sum of magnitudes <= 21.5 h^2: 86 u h^2 with one rounding per step (taken x 4: 344)
//   T, T +- delta      6 u h^2
//   the reference      d2 = fl-chain of the same sum: 5 u |c - q|^2 <= 125 u h^2 over the whole neighbourhood
//   => 569 u h^2 < delta = 1024 u h^2.  h >= r, and the band costs re-tests, never exactness.
#include "tnsx_kernels.h"
#include "tnsx_device.h"
#include "tnsx_pool.h"

#include <cfloat>

namespace tnsx {

typedef float v4f __attribute__((ext_vector_type(4)));

static constexpr int GQ = 16;                    // queries of a group (the N of the 16x16x4 MFMA)
static constexpr int G_CMAX = 480;               // candidate slots of a group (a multiple of 32)
static constexpr int G_LIST_CAP = 41;            // 16-bit entries of a lane's list
static constexpr int G_LIST_BYTES = 2 * G_LIST_CAP;   // 82: an odd number of 16-bit words, the lists of the four lanes of a query start in different banks
static constexpr int G_OFF_IDS = G_CMAX * 16;                        // A operands (16 B per slot)
static constexpr int G_OFF_LISTS = G_OFF_IDS + G_CMAX * 4;           // ids (4 B per slot)
static constexpr int G_OFF_XCH = G_OFF_LISTS + WAVE * G_LIST_BYTES;  // lists
static constexpr int G_LDS_BYTES = G_OFF_XCH + WAVE * 4;             // exchange words
static constexpr int G_OFF_STAGE = 0;                                // the staged block of records of a batch: in place of the A operands
static constexpr int G_STAGE_INTS = G_CMAX * 4;
static constexpr float G_DELTA_K = 1.0f / 16384.0f;                  // delta = 2^-14 h^2 (see above)
static constexpr float G_BIG = 3.0e38f;

// the reference's predicate arithmetic on scalars (TreeNSearch.cpp:2478-2483 / what GCC makes of it under the reference's flags)
template <int ARITH>
__device__ __forceinline__ float dist_sq1(float qx, float qy, float qz, float cx, float cy, float cz)
{
	const float dx = __fsub_rn(qx, cx), dy = __fsub_rn(qy, cy), dz = __fsub_rn(qz, cz);
	if (ARITH == 0) return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
	return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
}

// one result register of a tile: the lanes whose value passed (mask m) append `d` to their lists; bm = running max of appended values
#define TNSX_G_APPEND(m, S, d)                                                                                       \
	asm volatile("s_mov_b64 exec, %[mk]\n\t"                                                                         \
	             "ds_write_b16 %[ad], %[dv]\n\t"                                                                     \
	             "v_add_u32 %[ad], 2, %[ad]\n\t"                                                                     \
	             "v_max_f32 %[b], %[b], %[s]\n\t"                                                                    \
	             "s_mov_b64 exec, -1\n\t"                                                                            \
	             "v_add_u32 %[dv], 16, %[dv]"                                                                        \
	             : [ad] "+v"(addr), [b] "+v"(bm), [dv] "+v"(d)                                                       \
	             : [mk] "s"(m), [s] "v"(S))

template <int ARITH, bool SELF>
__global__ void __launch_bounds__(WAVE) k_query_groups(const QueryArgs a)
{
	if (a.abort_flag && *a.abort_flag != 0u) return;   // (see k_query_pool_fast)
	__shared__ __attribute__((aligned(16))) unsigned char lds[G_LDS_BYTES];
	const int lane = (int)threadIdx.x;
	const uint32_t j = (uint32_t)lane & 15u, qa = (uint32_t)lane >> 4;
	float4* const A = reinterpret_cast<float4*>(lds);
	uint32_t* const ids = reinterpret_cast<uint32_t*>(lds + G_OFF_IDS);
	uint32_t* const xch = reinterpret_cast<uint32_t*>(lds + G_OFF_XCH);
	int* const stage = reinterpret_cast<int*>(lds + G_OFF_STAGE);
	const uint32_t lds_base = (uint32_t)(uintptr_t)lds;   // (the low half of a generic LDS address is the LDS offset)
	const uint32_t list_base = lds_base + (uint32_t)G_OFF_LISTS + (uint32_t)lane * (uint32_t)G_LIST_BYTES;
	const uint32_t list_lim = list_base + 2u * (uint32_t)(G_LIST_CAP - 8);   // a round of two tiles appends at most 8 entries per lane

	const uint32_t n_occ = *a.n_occ_i;
	const uint32_t xcd = blockIdx.x & 7u, wx = blockIdx.x >> 3, n_wx = gridDim.x >> 3;
	const uint32_t c_lo = (uint32_t)(((uint64_t)n_occ * xcd) >> 3), c_hi = (uint32_t)(((uint64_t)n_occ * (xcd + 1u)) >> 3);
	const int nx = a.g.nx, ny = a.g.ny, nz = a.g.nz;
	const float inv_h = a.g.inv_h, h = 1.0f / inv_h;
	const float delta = G_DELTA_K * h * h, n_bound_c = 7.0f * h * h, n_bound_q = 0.8f * h * h;
	const float r2 = a.r2_fixed;

	PoolState ps = { 0u, 0u, 0u, 0u, 0u };
	uint32_t wave_hits = 0;
	uint2 rej = make_uint2(0u, 0u);
	uint32_t rej_n = 0;
	auto flush_rejects = [&]() {
		uint32_t hb = 0;
		if (lane == 0) hb = atomicAdd(a.n_heavy0, rej_n);
		hb = readfirstlane_u32(hb);
		if ((uint32_t)lane < rej_n) a.heavy0[hb + (uint32_t)lane] = rej;
		rej_n = 0;
	};
	auto pass_on = [&](uint2 oc) {   // this cell goes to the cell kernels (tnsx_query.hip)
		if ((uint32_t)lane == rej_n) rej = oc;
		if (++rej_n == (uint32_t)WAVE) flush_rejects();
	};

	// the 9 x 3 table entries of a cell's neighbourhood: lane = 7 * row + k (k < 3), row = 3 * (dz + 1) + (dy + 1)
	const int lk_r = lane / 7, lk_k = lane % 7;
	const int lk_dy = lk_r % 3 - 1, lk_dz = lk_r / 3 - 1;

	// every XCD owns one contiguous eighth of the (key-ordered) occupied-cell list; its waves take the cells of that eighth in turn
	uint32_t ci = c_lo + wx;
	uint2 oc_next = a.occ_i[ci < c_hi ? ci : c_lo < c_hi ? c_lo : 0u];
	for (; ci < c_hi; ci += n_wx) {
		const uint2 oc = make_uint2(readfirstlane_u32(oc_next.x), readfirstlane_u32(oc_next.y));
		{ const uint32_t cn = ci + n_wx; oc_next = a.occ_i[cn < c_hi ? cn : ci]; }   // (one cell ahead)
		const uint32_t key = oc.y;
		const int ix0 = (int)(key % (uint32_t)nx), iy0 = (int)((key / (uint32_t)nx) % (uint32_t)ny), iz0 = (int)(key / ((uint32_t)nx * (uint32_t)ny));
		const uint2 qr = a.table_i[key];
		const uint32_t q0 = readfirstlane_u32(qr.x), nq_all = readfirstlane_u32(qr.y) - q0;

		// ---- candidate rows: the three cells ix0 - 1 .. ix0 + 1 of the 9 rows, merged per row
		uint32_t rs[9], rl[9], slot0[9];
		uint32_t n_cand = 0;
		{
			const int cxk = ix0 - 1 + lk_k, cyk = iy0 + lk_dy, czk = iz0 + lk_dz;
			const bool use = lk_r < 9 && lk_k < 3 && cxk >= 0 && cxk < nx && cyk >= 0 && cyk < ny && czk >= 0 && czk < nz;
			const uint32_t idx = use ? ((uint32_t)czk * (uint32_t)ny + (uint32_t)cyk) * (uint32_t)nx + (uint32_t)cxk : 0u;
			const uint2 te = a.table_j[idx];
			const uint64_t nem = __builtin_amdgcn_ballot_w64(use && te.y > te.x);   // (empty entries may hold any (s, s))
			#pragma unroll
			for (int r = 0; r < 9; r++) {
				const uint32_t bits = (uint32_t)(nem >> (7 * r)) & 0x7u;
				uint32_t s = 0, e = 0;
				if (bits != 0u) {
					s = readlane_u32(te.x, 7 * r + __builtin_ctz(bits));
					e = readlane_u32(te.y, 7 * r + 31 - __builtin_clz(bits));
				}
				rs[r] = s; rl[r] = e - s; slot0[r] = n_cand;
				n_cand += e - s;
			}
		}
		if (n_cand > (uint32_t)G_CMAX || nq_all > (uint32_t)WAVE) { pass_on(oc); continue; }
		if (n_cand == 0u && a.shared_empty != 0u) continue;   // every offset already points at the shared empty record
		const uint32_t n_pad = (n_cand + 31u) & ~31u;

		// ---- the cell's query points (one per lane); those that get lists are a prefix (tnsx_set_query_count, see fast_query_loop)
		const uint32_t qsrc_all = q0 + ((uint32_t)lane < nq_all ? (uint32_t)lane : 0u);
		const uint32_t qorig_all = a.orig_i ? a.orig_i[qsrc_all] : __float_as_uint(a.xyzi_i[qsrc_all].w);
		const uint32_t nq = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((uint32_t)lane < nq_all && qorig_all < a.query_limit));
		if (nq == 0u) continue;

		// ---- local frame: origin = centre of the cell
		const float ocx = a.g.ox + ((float)ix0 + 0.5f) * h, ocy = a.g.oy + ((float)iy0 + 0.5f) * h, ocz = a.g.oz + ((float)iz0 + 0.5f) * h;
		// ---- batches of 16 queries.  The staged block of a batch's records takes the place of the A operands, so a cell with more than 16
		//      query points stages its candidates again (from the L1 / L2) for the next batch.
		uint32_t cell_hits = 0;
		bool failed = false;
		for (uint32_t b = 0; b < nq; b += (uint32_t)GQ) {
			// ---- candidates -> A operands + ids in LDS (all row loads issued back to back)
			float nmax = 0.0f;
			{
				float4 c[9];
				#pragma unroll
				for (int r = 0; r < 9; r++) {
					const uint32_t k = (uint32_t)lane < rl[r] ? (uint32_t)lane : 0u;
					c[r] = a.xyzi_j[rs[r] + k];
				}
				#pragma unroll
				for (int r = 0; r < 9; r++) {
					if ((uint32_t)lane < rl[r]) {
						const float lx = c[r].x - ocx, ly = c[r].y - ocy, lz = c[r].z - ocz;
						const float n2 = (lx * lx + ly * ly) + lz * lz;
						const uint32_t slot = slot0[r] + (uint32_t)lane;
						A[slot] = make_float4(-2.0f * lx, -2.0f * ly, -2.0f * lz, n2);
						ids[slot] = __float_as_uint(c[r].w);
						nmax = fmaxf(nmax, !(n2 <= n_bound_c) ? FLT_MAX : n2);
					}
				}
				#pragma unroll
				for (int r = 0; r < 9; r++) {
					for (uint32_t k0 = (uint32_t)WAVE; k0 < rl[r]; k0 += (uint32_t)WAVE) {   // (rows of more than 64 points: rare)
						const uint32_t k = k0 + (uint32_t)lane;
						if (k < rl[r]) {
							const float4 cc = a.xyzi_j[rs[r] + k];
							const float lx = cc.x - ocx, ly = cc.y - ocy, lz = cc.z - ocz;
							const float n2 = (lx * lx + ly * ly) + lz * lz;
							A[slot0[r] + k] = make_float4(-2.0f * lx, -2.0f * ly, -2.0f * lz, n2);
							ids[slot0[r] + k] = __float_as_uint(cc.w);
							nmax = fmaxf(nmax, !(n2 <= n_bound_c) ? FLT_MAX : n2);
						}
					}
				}
				if (n_cand + (uint32_t)lane < n_pad) A[n_cand + (uint32_t)lane] = make_float4(0.0f, 0.0f, 0.0f, G_BIG);
			}
			if (__builtin_amdgcn_ballot_w64(nmax == FLT_MAX) != 0ull) {   // a point far outside its cell (border cells of a trimmed grid): the bound does not hold
				wave_lds_fence();
				failed = true; break;
			}
			wave_lds_fence();

			const uint32_t qi = q0 + b + j;
			const bool is_query = b + j < nq;
			const uint32_t qsrc = is_query ? qi : q0;
			const float4 qv = a.xyzi_i[qsrc];
			const uint32_t qorig = a.orig_i ? a.orig_i[qsrc] : __float_as_uint(qv.w);
			const float lqx = qv.x - ocx, lqy = qv.y - ocy, lqz = qv.z - ocz;
			const float nq2 = (lqx * lqx + lqy * lqy) + lqz * lqz;
			if (__builtin_amdgcn_ballot_w64(is_query && !(nq2 <= n_bound_q)) != 0ull) { failed = true; break; }

			// ---- the tests
			const float bval = qa == 0u ? lqx : (qa == 1u ? lqy : (qa == 2u ? lqz : 1.0f));
			const float T = r2 - nq2;
			const float t_hi = is_query ? T + delta : -G_BIG, t_lo = T - delta;
			uint32_t addr = list_base;
			float bm = -G_BIG;
			uint32_t d0 = 4u * qa, d1 = d0 + 1u, d2 = d0 + 2u, d3 = d0 + 3u;
			bool overflow = false;
			{
				// software pipeline: the MFMAs of round t + 1 are issued (and the A operands of round t + 2 read) before the results of
				// round t are looked at; a round = two tiles of 16 candidates.  (Reads past n_pad stay inside the wave's LDS; their
				// products are never looked at.)
				const unsigned char* const a_rd = lds + (j * 4u + qa) * 4u;
				const v4f zero = { 0.0f, 0.0f, 0.0f, 0.0f };
#define TNSX_G_READ(t) a0 = *reinterpret_cast<const float*>(a_rd + (t) * 16u); a1 = *reinterpret_cast<const float*>(a_rd + (t) * 16u + 256u)
#define TNSX_G_ROUND(t, C0, C1, N0, N1)                                                                          \
				if (__builtin_amdgcn_ballot_w64(addr > list_lim) != 0ull) { overflow = true; break; }            \
				N0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bval, zero, 0, 0, 0);                              \
				N1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bval, zero, 0, 0, 0);                              \
				TNSX_G_READ((t) + 64u);                                                                          \
				m = __builtin_amdgcn_ballot_w64(C0[0] <= t_hi); TNSX_G_APPEND(m, C0[0], d0);                     \
				m = __builtin_amdgcn_ballot_w64(C0[1] <= t_hi); TNSX_G_APPEND(m, C0[1], d1);                     \
				m = __builtin_amdgcn_ballot_w64(C0[2] <= t_hi); TNSX_G_APPEND(m, C0[2], d2);                     \
				m = __builtin_amdgcn_ballot_w64(C0[3] <= t_hi); TNSX_G_APPEND(m, C0[3], d3);                     \
				m = __builtin_amdgcn_ballot_w64(C1[0] <= t_hi); TNSX_G_APPEND(m, C1[0], d0);                     \
				m = __builtin_amdgcn_ballot_w64(C1[1] <= t_hi); TNSX_G_APPEND(m, C1[1], d1);                     \
				m = __builtin_amdgcn_ballot_w64(C1[2] <= t_hi); TNSX_G_APPEND(m, C1[2], d2);                     \
				m = __builtin_amdgcn_ballot_w64(C1[3] <= t_hi); TNSX_G_APPEND(m, C1[3], d3)
				float a0, a1;
				uint64_t m;
				TNSX_G_READ(0u);
				v4f S0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bval, zero, 0, 0, 0);
				v4f S1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bval, zero, 0, 0, 0);
				v4f N0, N1;
				TNSX_G_READ(32u);
				for (uint32_t t = 0; t < n_pad; t += 64u) {
					TNSX_G_ROUND(t, S0, S1, N0, N1);
					if (t + 32u >= n_pad) break;
					TNSX_G_ROUND(t + 32u, N0, N1, S0, S1);
				}
#undef TNSX_G_ROUND
#undef TNSX_G_READ
			}
			asm volatile("" ::: "memory");
			wave_lds_fence();
			if (overflow) { failed = true; break; }

			// ---- lanes that appended a value inside the band [T - delta, T + delta]: the reference's own arithmetic decides.  The lane
			//      walks its list; the value is recomputed from the A operand in LDS (within delta of what the MFMA delivered, both
			//      being within 344 u h^2 of the real number), so an entry below T - 2 delta was appended below T - delta and stays;
			//      the others -- about one per flagged lane -- are tested on the original coordinates and dropped if they fail.
			const bool flagged = bm >= t_lo;
			if (__builtin_amdgcn_ballot_w64(flagged) != 0ull) {
				uint16_t* const lw = reinterpret_cast<uint16_t*>(lds + G_OFF_LISTS + lane * G_LIST_BYTES);
				const uint32_t n_e = (addr - list_base) >> 1;
				const float t_near = T - 2.0f * delta;
				uint32_t w = 0;
				for (uint32_t e = 0; __builtin_amdgcn_ballot_w64(flagged && e < n_e) != 0ull; e++) {
					const bool act = flagged && e < n_e;
					const uint32_t slot = act ? (uint32_t)lw[e] : 0u;
					const float4 av = A[slot];
					const float sp = ((av.w + av.x * lqx) + av.y * lqy) + av.z * lqz;
					const bool near = act && !(sp < t_near);
					bool keep = act;
					if (__builtin_amdgcn_ballot_w64(near) != 0ull) {
						uint32_t pos = rs[0] + slot;
						#pragma unroll
						for (int r = 1; r < 9; r++) pos = slot >= slot0[r] ? rs[r] + (slot - slot0[r]) : pos;
						const float4 cv = a.xyzi_j[near ? pos : rs[4]];
						const float dd = dist_sq1<ARITH>(qv.x, qv.y, qv.z, cv.x, cv.y, cv.z);
						if (near) keep = dd <= r2;
					}
					if (keep) { lw[w] = (uint16_t)slot; w++; }
				}
				if (flagged) addr = list_base + 2u * w;
				wave_lds_fence();
			}

			// ---- the block of records of this batch: [count, ids of the query's four lists] x 16
			const uint32_t n_l = (addr - list_base) >> 1;
			xch[lane] = n_l;
			wave_lds_fence();
			const uint32_t n0 = xch[j], n1 = xch[j + 16u], n2q = xch[j + 32u], n3 = xch[j + 48u];
			const uint32_t self_slot = (SELF && is_query) ? slot0[4] + (qi - rs[4]) : 0xffffffffu;   // row 4 = the cell's own row
			const uint32_t tot = n0 + n1 + n2q + n3 - ((SELF && is_query) ? 1u : 0u);
			const uint32_t len = is_query ? ((a.shared_empty != 0u && tot == 0u) ? 0u : tot + 1u) : 0u;
			uint32_t incl = len;
			incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
			incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
			incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
			incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
			const uint32_t excl = incl - len;
			const uint32_t block_len = readlane_u32(incl, 15);
			if (block_len == 0u) continue;
			if (block_len > (uint32_t)G_STAGE_INTS) { failed = true; break; }
			bool ok;
			const uint64_t off = pool_alloc<false>(a, ps, block_len, lane, ok);
			{
				const uint32_t self_q = (self_slot & 15u) >> 2;   // the quarter whose list holds the query itself
				uint32_t dst = excl + 1u + (qa > 0u ? n0 : 0u) + (qa > 1u ? n1 : 0u) + (qa > 2u ? n2q : 0u);
				if (SELF && is_query && qa > self_q) dst -= 1u;
				uint32_t* wr = reinterpret_cast<uint32_t*>(stage) + dst;
				const uint16_t* rd = reinterpret_cast<const uint16_t*>(lds + G_OFF_LISTS + lane * G_LIST_BYTES);
				const uint32_t n_copy = len != 0u ? n_l : 0u;
				for (uint32_t e = 0; __builtin_amdgcn_ballot_w64(e < n_copy) != 0ull; e++) {
					if (e < n_copy) {
						const uint32_t slot = rd[e];
						const uint32_t id = ids[slot];
						if (!(SELF && slot == self_slot)) { *wr = id; wr++; }
					}
				}
				if (qa == 0u && len != 0u) {
					stage[excl] = (int)tot;
					if (ok) a.offs_by_orig[qorig] = off + excl;
				}
			}
			wave_lds_fence();
			if (ok) {
				for (uint32_t f = (uint32_t)lane; f < block_len; f += (uint32_t)WAVE) a.records[off + f] = stage[f];
			}
			cell_hits += block_len - (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(qa == 0u && len != 0u));
			wave_lds_fence();
		}
		if (failed) { pass_on(oc); wave_lds_fence(); continue; }   // (records of earlier batches stay behind as holes; the cell kernels write the cell again)
		wave_hits += cell_hits;
	}
	if (rej_n) flush_rejects();
	pool_wave_done<false>(a, ps, wave_hits, lane);
}

template <int ARITH, bool SELF>
static void launch_groups_t(const QueryArgs& a, int n_cus, int waves_per_cu, hipStream_t s)
{
	const int blocks = ((n_cus * waves_per_cu + 7) / 8) * 8;   // a multiple of 8: every XCD gets the same number (workgroup b runs on XCD b % 8)
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_groups<ARITH, SELF>), dim3(blocks), dim3(WAVE), 0, s, a);
}

void launch_query_groups(const QueryArgs& a, const QueryConfig& c, int n_compute_units, hipStream_t s)
{
	const int wpc = (c.group_waves_per_cu >= 1 && c.group_waves_per_cu <= 32) ? c.group_waves_per_cu : 10;
	if (c.arith == 0) { if (c.self) launch_groups_t<0, true>(a, n_compute_units, wpc, s); else launch_groups_t<0, false>(a, n_compute_units, wpc, s); }
	else { if (c.self) launch_groups_t<1, true>(a, n_compute_units, wpc, s); else launch_groups_t<1, false>(a, n_compute_units, wpc, s); }
	// what it passed on: the three cell tiers over its worklist
	QueryArgs t = a;
	t.occ_i = a.heavy0;
	t.n_occ_i = a.n_heavy0;
	QueryConfig ct = c;
	ct.groups = false;
	launch_query(t, ct, n_compute_units, s);
}

}  // namespace tnsx
