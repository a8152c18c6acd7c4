// gfx950 kernels: THE QUERY (27-cell distance test; single-pass pool mode and count / fill passes).
//
// Query design (one wave64 per occupied cell of the query set; no MFMA; LDS only as per-wave staging: the deal table of a cell's candidates and the
// block of its records, see "Candidate dealing through an LDS table" and "Whole-cell staging" below):
//   * the 27 neighbour cells of the candidate set are looked up by 27 lanes in one round trip and merged into 9
//     x-contiguous runs of the sorted candidate array (row-major keys: x-neighbours are adjacent in sorted order);
//   * the concatenation of the 9 runs is dealt to the lanes slot by slot (slot = chunk*64 + lane) through a table of sorted positions that every
//     run writes into the wave's LDS staging area once per cell (the general kernel: 8 scalar-operand compares per slot), so ALL candidate loads
//     (one coalesced 16-byte load per lane and chunk) are issued back to back and land directly in registers -- one memory round trip per cell;
//   * the occupied-cell list entry is prefetched two cells ahead and the 27 lookups one cell ahead, so that the only
//     exposed latency per cell is the candidate load, which the other resident waves hide;
//   * the cell's query points are broadcast one at a time (v_readlane); each is tested against all register-resident
//     candidates, two chunks per packed-fp32 instruction (v_pk_add/mul/fma_f32), and the hits are compacted with
//     ballot + mbcnt into the block of the cell's records in LDS, which leaves with full-wave non-temporal stores (pool mode), or straight into
//     the query's CSR record (fill), or are just counted (count);
//   * work assignment is XCD-aware: workgroup b runs on XCD b % 8, and every XCD owns one contiguous eighth of the
//     (roughly key-ordered) occupied-cell list, so the three z-planes a wave touches stay in that XCD's L2.
// The distance arithmetic is spelled op by op (file compiled with -ffp-contract=off) and is bit-identical to the
// reference's AVX2 path / BruteforceNSearch in either arithmetic mode (TreeNSearch.cpp:2478-2486, BF.cpp:88).
#include "tnsx_kernels.h"
#include "tnsx_device.h"
#include "tnsx_pool.h"

#include <cfloat>
#include <cstdlib>

namespace tnsx {

// =====================================================================================================
// the query
// =====================================================================================================
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#ifndef TNSX_CULL_FROM
#define TNSX_CULL_FROM 448   // fixed radius: cells with more candidates than this are culled first (measured: 512 / 448 / 384 -> C2 1.775 / 1.766 / 2.003 ms,
                             // C3 2.54 / 2.44 / 2.61 ms, C4 at 20 M 5.77 / 5.68 / 5.58 ms)
#endif
#ifndef TNSX_CULL_FROM_VARIABLE
#define TNSX_CULL_FROM_VARIABLE 320   // per-point radii (a cell edge of r_max, most radii well below it: the cull removes more): round 3, with the LDS deal table,
                                      // 0 / 320 / 448 -> C4 at 10 M 2.54 / 2.44 / 2.50 ms (profiles/r3_query_ab_cull_threshold.txt); fixed radius: 1.86 / 1.87 / 1.58
#endif
#ifndef TNSX_FAT_CULL
#define TNSX_FAT_CULL 1   // the second tier repeats the cull (its cells are those of which > 512 candidates survive) and loops over the survivors only
#endif
#ifndef TNSX_LANE_OPAQUE
#define TNSX_LANE_OPAQUE 1   // round 5: lane-derived constants of the cell bodies are recomputed per cell instead of living in (and spilling from) a dozen VGPRs
#endif
#ifndef TNSX_TOTAL_BY_DPP
#define TNSX_TOTAL_BY_DPP 1
#endif
#ifndef TNSX_CULL
#define TNSX_CULL 1   // first tier: cells with 513..1024 candidates are culled against the bounding box of their query points (fast_cell_culled)
#endif


static constexpr int Q_THREADS = 256;
static constexpr int Q_WAVES = Q_THREADS / WAVE;
static constexpr int Q_MAXPAIRS = 4;                       // chunk pairs per batch
static constexpr int Q_SLOTS = Q_MAXPAIRS * 2 * WAVE;      // 512 candidates per batch

// squared distances of one query to two candidates at once (packed fp32, every op individually rounded)
template <int ARITH>
__device__ __forceinline__ v2f dist_sq2(float qx, float qy, float qz, v2f cx, v2f cy, v2f cz)
{
	const v2f dx = (v2f)(qx) - cx;
	const v2f dy = (v2f)(qy) - cy;
	const v2f dz = (v2f)(qz) - cz;
	if (ARITH == 0) {
		return (dx * dx + dy * dy) + dz * dz;                                                   // STRICT
	}
	else {
		return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));   // CONTRACTED
	}
}

// max of a wave-uniform and a per-lane value as ONE v_max_f32 (fmaxf() comes with a canonicalising v_max x, x per operand in IEEE
// mode; neither operand is ever a NaN here: squared radii, or -1 in padding lanes)
__device__ __forceinline__ float max_raw(float uniform, float v)
{
	float r;
	asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(v));
	return r;
}

// Wave-uniform description of the 9 merged candidate runs of one cell.  Deliberately NINE NAMED SCALARS per field and
// not arrays: with arrays the compiler turns the select chain below into a table lookup and parks the table in LDS.
struct Runs {
	uint32_t p1, p2, p3, p4, p5, p6, p7, p8;        // first slot of run r (run 0 starts at slot 0)
	uint32_t d0, d1, d2, d3, d4, d5, d6, d7, d8;    // sorted position = slot + d_r
	uint32_t total;
	uint32_t p9, d9;                                // tenth run (own-cell-first order of the fast kernels, see extract_runs_own_first)
};

// What stays live across the query loop: the two VGPRs the 18 scalars are extracted from, plus two scalars.
struct RunRef {
	uint32_t run_start, run_len;   // per lane; lanes 0,3,..,24 hold run 0..8
	uint32_t total;                // wave-uniform: number of candidates
};

__device__ __forceinline__ Runs extract_runs(uint32_t run_start, uint32_t run_len)
{
	Runs R;
	uint32_t acc = 0, rs, rn, p0_unused;
#define TNSX_RUN(r, P, D)                                                       \
	rs = readlane_u32(run_start, 3 * r); rn = readlane_u32(run_len, 3 * r); \
	P = acc; D = rs - acc; acc += rn;
	// slot order: the CENTRE row (physical run 4, it contains the query cell itself) first, then the other eight.  The query
	// cell's own points therefore sit at slots [len(cell x-1), len(cell x-1) + nq), i.e. in chunk 0 or 1 for simple cells.
	TNSX_RUN(4, p0_unused, R.d0) TNSX_RUN(0, R.p1, R.d1) TNSX_RUN(1, R.p2, R.d2) TNSX_RUN(2, R.p3, R.d3) TNSX_RUN(3, R.p4, R.d4)
	TNSX_RUN(5, R.p5, R.d5) TNSX_RUN(6, R.p6, R.d6) TNSX_RUN(7, R.p7, R.d7) TNSX_RUN(8, R.p8, R.d8)
#undef TNSX_RUN
	(void)p0_unused;
	R.total = acc;
	return R;
}

__device__ __forceinline__ uint32_t slot_to_src(uint32_t slot, const Runs R)
{
	uint32_t d = R.d0;
	d = slot >= R.p1 ? R.d1 : d;
	d = slot >= R.p2 ? R.d2 : d;
	d = slot >= R.p3 ? R.d3 : d;
	d = slot >= R.p4 ? R.d4 : d;
	d = slot >= R.p5 ? R.d5 : d;
	d = slot >= R.p6 ? R.d6 : d;
	d = slot >= R.p7 ? R.d7 : d;
	d = slot >= R.p8 ? R.d8 : d;
	return slot + d;
}

// Slot order of the fast kernels when the query set is the candidate set: the query cell's OWN points come
// first, so query t of the cell is candidate slot t -- always chunk 0, bit t -- and self exclusion is one scalar bit-clear
// per query with no state.  The centre row [x-1 | own | x+1] (contiguous in sorted order) is split into
// run 0 = [own | x+1] and run 1 = [x-1]; the other eight rows follow: ten runs, nine selects per slot.
// (measured against the centre-row-first order with a self bit that is shifted along: C2 -8.6 %, C3 -5 %, C4 -2.5 % on the query)
__device__ __forceinline__ Runs extract_runs_own_first(uint32_t run_start, uint32_t run_len, uint32_t q_first)
{
	Runs R;
	uint32_t acc, rs, rn;
	rs = readlane_u32(run_start, 12); rn = readlane_u32(run_len, 12);   // centre row
	R.d0 = q_first; acc = rs + rn - q_first;                             // [own | x+1]
	R.p1 = acc; R.d1 = rs - acc; acc += q_first - rs;                    // [x-1]
#define TNSX_RUN(r, P, D)                                                       \
	rs = readlane_u32(run_start, 3 * r); rn = readlane_u32(run_len, 3 * r); \
	P = acc; D = rs - acc; acc += rn;
	TNSX_RUN(0, R.p2, R.d2) TNSX_RUN(1, R.p3, R.d3) TNSX_RUN(2, R.p4, R.d4) TNSX_RUN(3, R.p5, R.d5)
	TNSX_RUN(5, R.p6, R.d6) TNSX_RUN(6, R.p7, R.d7) TNSX_RUN(7, R.p8, R.d8) TNSX_RUN(8, R.p9, R.d9)
#undef TNSX_RUN
	R.total = acc;
	return R;
}
template <bool OWN_FIRST>
__device__ __forceinline__ Runs extract_runs_t(uint32_t run_start, uint32_t run_len, uint32_t q_first)
{
	if (OWN_FIRST) return extract_runs_own_first(run_start, run_len, q_first);
	Runs R = extract_runs(run_start, run_len);
	R.p9 = R.total; R.d9 = 0;
	return R;
}
// ---------------------------------------------------------------------------------------------------------------------
// Candidate dealing through an LDS table (round 3).  A select chain like slot_to_src above costs nine compare + select pairs per slot and chunk -- all of
// them instructions that touch an SGPR or VCC, 4.3 cycles each: 77 cycles per chunk, 22 % of the kernel's vector instructions
// (profiles/r2_c2_pmc.json).  Instead every run writes the sorted positions of its candidates into a table in the wave's staging
// area ONCE per cell: run r covers the slots [p_r, p_r+1), lane l of the piece writes p_r + d_r + l at table[p_r + l] with
// ds_write_addtid_b32 (address = M0 + 4 * lane: no address register, no per-lane address arithmetic; the run's extent is the exec
// mask).  One scalar-operand add and one LDS store per run instead of 18 vector instructions per chunk; a chunk then reads its 64
// table entries with one ds_read.  The table lives where the cell's records are staged afterwards (the staging area is empty while
// the candidates are loaded; LDS operations of a wave execute in order).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void deal_run(uint32_t tbl, uint32_t p0, uint32_t p1, uint32_t d, uint32_t lane)
{
	uint32_t n = p1 - p0;   // wave-uniform
	uint32_t pos = p0;
	if (n - 1u < 64u) {     // the usual case, without the loop's bookkeeping: one piece (a run is three cells of the grid)
		const uint32_t v = lane + (pos + d);
		asm volatile("s_mov_b32 m0, %[b]\n\ts_lshr_b64 exec, -1, %[sh]\n\tds_write_addtid_b32 %[v]\n\ts_mov_b64 exec, -1"
		             : : [b] "s"(tbl + 4u * pos), [sh] "s"(64u - n), [v] "v"(v) : "memory", "scc");   // (s_lshr_b64 writes SCC: round 6 found the clobber missing)
		return;
	}
	while (n != 0u) {
		const uint32_t c = n < 64u ? n : 64u;
		const uint32_t v = lane + (pos + d);
		asm volatile("s_mov_b32 m0, %[b]\n\ts_lshr_b64 exec, -1, %[sh]\n\tds_write_addtid_b32 %[v]\n\ts_mov_b64 exec, -1"
		             : : [b] "s"(tbl + 4u * pos), [sh] "s"(64u - c), [v] "v"(v) : "memory", "scc");
		n -= c; pos += c;
	}
}
template <bool OWN_FIRST>
__device__ __forceinline__ void deal_table(uint32_t tbl, const Runs R, uint32_t lane)
{
	deal_run(tbl, 0u, R.p1, R.d0, lane);
	deal_run(tbl, R.p1, R.p2, R.d1, lane);
	deal_run(tbl, R.p2, R.p3, R.d2, lane);
	deal_run(tbl, R.p3, R.p4, R.d3, lane);
	deal_run(tbl, R.p4, R.p5, R.d4, lane);
	deal_run(tbl, R.p5, R.p6, R.d5, lane);
	deal_run(tbl, R.p6, R.p7, R.d6, lane);
	deal_run(tbl, R.p7, R.p8, R.d7, lane);
	deal_run(tbl, R.p8, R.p9, R.d8, lane);
	if (OWN_FIRST) deal_run(tbl, R.p9, R.total, R.d9, lane);
}

// Round 6: extraction of the runs and dealing in ONE pass for the first tier (the scalar unit is shared by the four SIMDs of a CU and this kernel issues ~40 scalar
// instructions per query; extract_runs_t + deal_table above were ~170 of a cell's ~500).  A run's table entries are `first sorted position of the run + lane`, its
// slots are consecutive: M0 walks through the table (s_lshl2_add_u32 m0, n, m0), the run's extent is the exec mask (s_bfm_b64), one v_add and one
// ds_write_addtid_b32 per run -- four scalar instructions per run instead of fourteen.  Runs of 64 slots and more (three cells of a row holding > 63 points: a few
// cells in a thousand at C2) make the caller take the general way.  -> p1 = slots of run 0 (the centre run: kept whole by the cull), d0 = sorted position of slot 0.
#ifndef TNSX_DEAL_DIRECT
#define TNSX_DEAL_DIRECT 1
#endif
template <bool OWN_FIRST>
__device__ __forceinline__ bool deal_runs_direct(uint32_t tbl, const RunRef RR, uint32_t q_first, uint32_t lane, uint32_t& p1, uint32_t& d0)
{
	// (lanes 0, 3, ..., 24 hold the nine rows; every other lane's run_len is not a run)
	const uint64_t long_run = __builtin_amdgcn_ballot_w64(RR.run_len > 63u) & 0x1249249ull;
	if (long_run != 0ull) return false;
	uint32_t rs[10], rn[10];
	const uint32_t cs = readlane_u32(RR.run_start, 12), cn = readlane_u32(RR.run_len, 12);   // the centre row
	if (OWN_FIRST) {
		rs[0] = q_first; rn[0] = cs + cn - q_first;      // [own | x + 1]
		rs[1] = cs; rn[1] = q_first - cs;                // [x - 1]
	}
	else { rs[0] = cs; rn[0] = cn; rs[1] = cs; rn[1] = 0u; }
	#pragma unroll
	for (int r = 0; r < 8; r++) {
		const int src = r < 4 ? r : r + 1;               // rows 0..3, 5..8
		rs[2 + r] = readlane_u32(RR.run_start, 3 * src); rn[2 + r] = readlane_u32(RR.run_len, 3 * src);
	}
	p1 = rn[0]; d0 = rs[0];
	uint32_t t0, t1;
#define TNSX_DEAL_PIECE(K, T) "s_bfm_b64 exec, %[n" #K "], 0\n\tv_add_u32 %[" #T "], %[s" #K "], %[lane]\n\tds_write_addtid_b32 %[" #T "]\n\ts_lshl2_add_u32 m0, %[n" #K "], m0\n\t"
	asm volatile("s_mov_b32 m0, %[tbl]\n\t"
	             TNSX_DEAL_PIECE(0, t0) TNSX_DEAL_PIECE(1, t1) TNSX_DEAL_PIECE(2, t0) TNSX_DEAL_PIECE(3, t1) TNSX_DEAL_PIECE(4, t0)
	             TNSX_DEAL_PIECE(5, t1) TNSX_DEAL_PIECE(6, t0) TNSX_DEAL_PIECE(7, t1) TNSX_DEAL_PIECE(8, t0) TNSX_DEAL_PIECE(9, t1)
	             "s_mov_b64 exec, -1"
	             : [t0] "=&v"(t0), [t1] "=&v"(t1)
	             : [tbl] "s"(tbl), [lane] "v"(lane),
	               [s0] "s"(rs[0]), [n0] "s"(rn[0]), [s1] "s"(rs[1]), [n1] "s"(rn[1]), [s2] "s"(rs[2]), [n2] "s"(rn[2]), [s3] "s"(rs[3]), [n3] "s"(rn[3]),
	               [s4] "s"(rs[4]), [n4] "s"(rn[4]), [s5] "s"(rs[5]), [n5] "s"(rn[5]), [s6] "s"(rs[6]), [n6] "s"(rn[6]), [s7] "s"(rs[7]), [n7] "s"(rn[7]),
	               [s8] "s"(rs[8]), [n8] "s"(rn[8]), [s9] "s"(rs[9]), [n9] "s"(rn[9])
	             : "memory", "scc");   // (s_lshl2_add_u32 writes SCC)
#undef TNSX_DEAL_PIECE
	return true;
}

// 27 neighbour lookups of the cell with this key (lanes 0..26), wave-uniform key
template <bool SPARSE = false>
__device__ __forceinline__ void lookup_cell(const QueryArgs& a, uint32_t key, bool valid, int lane, uint32_t& s, uint32_t& e)
{
	s = 0; e = 0;
	const uint32_t nx = (uint32_t)a.g.nx, ny = (uint32_t)a.g.ny, nz = (uint32_t)a.g.nz;
	const int cx = (int)(key % nx);
	const int cy = (int)((key / nx) % ny);
	const int cz = (int)(key / (nx * ny));
	// branch-free (a load inside an exec region would be waited for inside it): lanes without a neighbour cell read entry 0
	const int x = cx + (lane % 3) - 1, y = cy + ((lane / 3) % 3) - 1, z = cz + (lane / 9) - 1;
	const bool use = valid && lane < 27 && x >= 0 && x < (int)nx && y >= 0 && y < (int)ny && z >= 0 && z < (int)nz;
	const uint32_t idx = use ? ((uint32_t)z * ny + (uint32_t)y) * nx + (uint32_t)x : 0u;   // (the dense table has at most 2^30 cells)
	if (SPARSE) {
		// sparse grid: the cell's entry of the key-ordered occupied-cell list, through the block index (three or four dependent loads per lane)
		uint2 r = make_uint2(0u, 0u);
		if (use) r = sparse_find(a.socc_j, a.blk_j, a.sparse_shift, idx);
		s = r.x; e = r.y;
		return;
	}
	const uint2 r = a.table_j[idx];
	s = use ? r.x : 0u;
	e = use ? r.y : 0u;
}
// The same for the fast kernels, with the lane's three offsets packed into one register (dx | dy << 2 | dz << 4, each 0..2; lanes >= 27: 63): the compiler keeps
// lane % 3, lane / 3 % 3 and lane / 9 of the version above in three VGPRs for the life of the kernel, and at five waves per SIMD that is what spills in the
// instantiations with per-point radii (round 5).  The caller makes `pack` opaque per cell, so that it is unpacked (three v_bfe) where it is used.
__device__ __forceinline__ uint32_t neighbour_pack(int lane) { return lane < 27 ? (uint32_t)(lane % 3) | ((uint32_t)((lane / 3) % 3) << 2) | ((uint32_t)(lane / 9) << 4) : 63u; }
__device__ __forceinline__ void lookup_cell_packed(const QueryArgs& a, uint32_t key, bool valid, uint32_t pack, uint32_t& s, uint32_t& e)
{
	const uint32_t nx = (uint32_t)a.g.nx, ny = (uint32_t)a.g.ny, nz = (uint32_t)a.g.nz;
	const int cx = (int)(key % nx);
	const int cy = (int)((key / nx) % ny);
	const int cz = (int)(key / (nx * ny));
	const int x = cx + (int)(pack & 3u) - 1, y = cy + (int)((pack >> 2) & 3u) - 1, z = cz + (int)(pack >> 4) - 1;   // (pack == 63: z = cz + 2, never used)
	const bool use = valid && pack != 63u && x >= 0 && x < (int)nx && y >= 0 && y < (int)ny && z >= 0 && z < (int)nz;
	const uint32_t idx = use ? ((uint32_t)z * ny + (uint32_t)y) * nx + (uint32_t)x : 0u;
	const uint2 r = a.table_j[idx];
	s = use ? r.x : 0u;
	e = use ? r.y : 0u;
}
// {first, one past last} sorted position of the query cell itself (wave-uniform key)
template <bool SPARSE = false>
__device__ __forceinline__ uint2 own_range(const QueryArgs& a, uint32_t key)
{
	if (SPARSE) return sparse_find(a.socc_i, a.blk_i, a.sparse_shift, key);
	return a.table_i[key];
}

enum { MODE_COUNT = 0, MODE_FILL = 1, MODE_POOL = 2 };

// (the per-wave bump allocator over the record pool: tnsx_pool.h)
// Appends the set lanes of mask m (their value v) to rec[1 + pos...] in lane order (rec[0] is the record's count word).  Hand-scheduled: exec is loaded from
// the mask and restored to all-ones (every lane of the wave is active wherever this is called), which costs two
// scalar instructions instead of the compiler's s_and_saveexec / s_cbranch_execz / s_or triple -- the scalar unit is
// the scarce resource of this kernel.  The store is younger than every load the compiler tracks, so its untracked
// vmcnt increment can only make the compiler's waits longer, never shorter.
__device__ __forceinline__ void emit_chunk(const int* rec, uint32_t pos, uint64_t m, uint32_t v)
{
	uint32_t tmp;
	asm volatile(
		"s_mov_b64 exec, %[m]\n\t"
		"v_mbcnt_lo_u32_b32 %[t], %[mlo], 0\n\t"
		"v_mbcnt_hi_u32_b32 %[t], %[mhi], %[t]\n\t"
		"v_add_lshl_u32 %[t], %[t], %[pos], 2\n\t"
		"global_store_dword %[t], %[v], %[base] offset:4\n\t"   // (nt / sc1 stores were tried: 1.79 -> 2.45..2.70 ms, the L2 must merge these partial lines)
		"s_mov_b64 exec, -1"
		: [t] "=&v"(tmp)
		: [m] "s"(m), [mlo] "s"((uint32_t)m), [mhi] "s"((uint32_t)(m >> 32)), [pos] "s"(pos), [v] "v"(v), [base] "s"(rec)
		: "memory");
}

// V# of the record storage (index-addressed buffer stores of the fast kernels)
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i record_rsrc(const int* base)
{
	// gfx9 buffer resource: base address [47:0], stride [61:48] = 4 bytes, num_records = 2^32 - 1 (never out of range),
	// word 3 = DATA_FORMAT_32 (0x00020000, as used by composable_kernel for gfx9)
	const uint64_t b = (uint64_t)base;
	v4i r;
	r.x = (int)(uint32_t)b;
	r.y = (int)(((uint32_t)(b >> 32) & 0xffffu) | (4u << 16));
	r.z = -1;
	r.w = 0x00020000;
	return r;
}
// ---------------------------------------------------------------------------------------------------------------------
// LDS-staged emission.  Measured (profiles/r2_emit_ab.txt): the sparse global stores of the hit compaction, not
// instruction issue, are what the fast kernel waits for -- ~6 store instructions per query with ~10 live lanes each keep the
// texture-address FIFO full (SQ_VMEM_TA_ADDR_FIFO_FULL 94 % of busy cycles), and spreading a chunk's hits over the whole
// record (lane-major order in global memory) made it worse.  So the hits of a query are compacted into the wave's LDS staging
// area first, lane-major (P[l] from one v_mbcnt chain, then exec = mask: ds_write, address += 4), and the record leaves
// with ONE full-wave coalesced store per 64 indices.  The count is what lane 63's address ends up at.
// ---------------------------------------------------------------------------------------------------------------------
template <int SLOTS>
__device__ __forceinline__ uint32_t* record_stage()
{
	__shared__ uint32_t s_stage[Q_WAVES * SLOTS];
	return s_stage + readfirstlane_u32(threadIdx.x / WAVE) * SLOTS;
}
#define TNSX_LDS_IN(K, M, V) [m##K] "s"(M), [v##K] "v"(V)

// One batch of <= NC*64 candidates (register resident) against the nq query points held one per lane in qv.
//   MODE_COUNT: run_cnt (lane t) += hits of query t
//   MODE_FILL : record of query t starts at my_off (lane t); indices appended at my_off + 1 + run_cnt
//   MODE_POOL : record allocated here (single-batch cells only); my_off (lane t) receives its offset
// SELF: 0 = other set, 1 = exclude the query itself with an index compare in every chunk (the fast kernels clear one mask bit instead)
template <int ARITH, bool VARIABLE, bool SYM, int SELF, int MODE, int NC, bool EXACT_NC>
__device__ __forceinline__ void process_batch(const QueryArgs& a, const RunRef RR, uint32_t wb, int lane, const float4& qv, float qr2, uint32_t qb,
                                              uint32_t nq, uint64_t& my_off, uint32_t& run_cnt, PoolState& ps, uint32_t& wave_hits)
{
	constexpr int NP = (NC + 1) / 2;
	// the 18 run scalars live only during the load phase (SGPR pressure); the query loop needs just d4 and total
	const Runs R = extract_runs(RR.run_start, RR.run_len);
	// ---- candidates of this batch -> registers (all loads independent, issued back to back)
	//      Branch-free on purpose: a load inside `if (slot < total)` gets its own exec region and its own
	//      s_waitcnt, which serialises the round trips.  Out-of-range slots read a clamped (valid) address instead and
	//      are overwritten with padding afterwards.
	// chunks paired as (x_k, x_k+1) so that one packed instruction serves two chunks
	v2f cx[NP], cy[NP], cz[NP];
	uint32_t cid[2 * NP];
	float cr2[2 * NP];
	float4 craw[2 * NP];
	float r2raw[2 * NP];
	#pragma unroll
	for (int k = 0; k < 2 * NP; k++) {
		if (k < NC) {
			const uint32_t slot = wb + (uint32_t)(k * WAVE + lane);
			const uint32_t src = slot < R.total ? slot_to_src(slot, R) : R.d0;   // R.d0 = first candidate of the cell
			craw[k] = a.xyzi_j[src];
			if (SYM) r2raw[k] = a.r2_j[src];
		}
	}
	#pragma unroll
	for (int k = 0; k < 2 * NP; k++) {
		float4 c = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, __uint_as_float(0xffffffffu));
		float r2c = -1.0f;
		if (k < NC) {
			const uint32_t slot = wb + (uint32_t)(k * WAVE + lane);
			const bool valid = slot < R.total;
			c.x = valid ? craw[k].x : FLT_MAX; c.y = valid ? craw[k].y : FLT_MAX; c.z = valid ? craw[k].z : FLT_MAX;
			c.w = valid ? craw[k].w : __uint_as_float(0xffffffffu);
			if (SYM) r2c = valid ? r2raw[k] : -1.0f;
		}
		cx[k >> 1][k & 1] = c.x; cy[k >> 1][k & 1] = c.y; cz[k >> 1][k & 1] = c.z;
		cid[k] = __float_as_uint(c.w);
		cr2[k] = r2c;
	}
	// ---- every query of the cell against them
	for (uint32_t t = 0; t < nq; t++) {
		const float qx = readlane_f32(qv.x, (int)t), qy = readlane_f32(qv.y, (int)t), qz = readlane_f32(qv.z, (int)t);
		const float r2 = VARIABLE ? readlane_f32(qr2, (int)t) : a.r2_fixed;
		const uint32_t qi = readlane_u32(__float_as_uint(qv.w), (int)t);
		// hit masks of all chunks first: every compare writes its lane mask to an SGPR pair, the rest is scalar work
		uint64_t m[NC];
		#pragma unroll
		for (int h = 0; h < NP; h++) {
			const v2f d2 = dist_sq2<ARITH>(qx, qy, qz, cx[h], cy[h], cz[h]);
			#pragma unroll
			for (int u = 0; u < 2; u++) {
				const int k = 2 * h + u;
				if (k < NC) {
					// symmetric search: d2 <= r_i^2 || d2 <= r_j^2 is d2 <= max(r_i^2, r_j^2) -- one v_max instead of a second compare
					// and a scalar or (padding lanes carry r_j^2 = -1 and d2 = +inf)
					m[k] = __builtin_amdgcn_ballot_w64(d2[u] <= (SYM ? max_raw(r2, cr2[k]) : r2));
					// self exclusion by index: one VALU compare + one s_and per chunk.  (Clearing the one self bit with
					// scalar ops costs 4-5 SALU per chunk, and the CU's single scalar unit is as scarce as its 4 SIMDs.)
					if (SELF == 1) m[k] &= __builtin_amdgcn_ballot_w64(cid[k] != qi);
				}
			}
		}
		uint32_t cnt = 0;
		#pragma unroll
		for (int k = 0; k < NC; k++) cnt += (uint32_t)__popcll(m[k]);

		if (MODE != MODE_COUNT) {
			uint64_t off;
			bool ok = true;
			if (MODE == MODE_POOL) {
				off = pool_alloc<true>(a, ps, cnt + 1u, lane, ok);
				if ((uint32_t)lane == t) my_off = ok ? off : ~0ull;
			}
			else {
				const uint32_t lo = readlane_u32((uint32_t)my_off, (int)t), hi = readlane_u32((uint32_t)(my_off >> 32), (int)t);
				off = (((uint64_t)hi << 32) | lo) + readlane_u32(run_cnt, (int)t);
			}
			if (ok) {
				const int* dst = a.records + off;   // record start; emit_chunk skips the count word
				uint32_t pos = 0;
				#pragma unroll
				for (int k = 0; k < NC; k++) {
					emit_chunk(dst, pos, m[k], cid[k]);
					pos += (uint32_t)__popcll(m[k]);
				}
			}
		}
		if ((uint32_t)lane == t) run_cnt += cnt;
		wave_hits += cnt;
	}
}

// FULL = false: only the 8-chunk body is instantiated (used on the rare multi-batch path to keep the code small)
template <int ARITH, bool VARIABLE, bool SYM, int SELF, int MODE, bool FULL>
__device__ __forceinline__ void process_batch_nc(const QueryArgs& a, const RunRef RR, uint32_t wb, uint32_t nb, int lane, const float4& qv, float qr2,
                                                 uint32_t qb, uint32_t nq, uint64_t& my_off, uint32_t& run_cnt, PoolState& ps, uint32_t& wave_hits)
{
	const uint32_t nc = FULL ? (nb + WAVE - 1) / WAVE : 8u;
	switch (nc) {
	case 1: process_batch<ARITH, VARIABLE, SYM, SELF, MODE, 1, FULL>(a, RR, wb, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, wave_hits); break;
	case 2: process_batch<ARITH, VARIABLE, SYM, SELF, MODE, 2, FULL>(a, RR, wb, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, wave_hits); break;
	case 3: process_batch<ARITH, VARIABLE, SYM, SELF, MODE, 3, FULL>(a, RR, wb, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, wave_hits); break;
	case 4: process_batch<ARITH, VARIABLE, SYM, SELF, MODE, 4, FULL>(a, RR, wb, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, wave_hits); break;
	case 5: process_batch<ARITH, VARIABLE, SYM, SELF, MODE, 5, FULL>(a, RR, wb, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, wave_hits); break;
	case 6: process_batch<ARITH, VARIABLE, SYM, SELF, MODE, 6, FULL>(a, RR, wb, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, wave_hits); break;
	case 7: process_batch<ARITH, VARIABLE, SYM, SELF, MODE, 7, FULL>(a, RR, wb, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, wave_hits); break;
	default: process_batch<ARITH, VARIABLE, SYM, SELF, MODE, 8, FULL>(a, RR, wb, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, wave_hits); break;
	}
}

// SPARSE: the cells of a grid without a dense table (tnsx_build.hip "SPARSE grids"): look-ups through the block index.  Its own instantiation of the general
// kernel -- the only kernel that serves such grids -- so that the dense kernels carry none of it (as a run-time branch it cost the first tier two spilled
// VGPRs; instantiating the fast tiers too doubles the compile time for a query 15 % faster: measured, not kept)
template <int ARITH, bool VARIABLE, bool SYM, bool SELF, int MODE, bool SPARSE = false>
__global__ void __launch_bounds__(Q_THREADS) k_query(const QueryArgs a)
{
	if (a.abort_flag && *a.abort_flag != 0u) return;   // (this attempt is already known to be wrong; its sorted arrays may have holes)
	const int lane = lane_id();
	const uint32_t w = readfirstlane_u32(threadIdx.x / WAVE);
	const uint32_t n_occ = *a.n_occ_i;
	// XCD-aware assignment: this workgroup's XCD owns the contiguous cell range [lo, hi)
	const uint32_t xcd = blockIdx.x & 7u;
	const uint32_t lo = (uint32_t)(((uint64_t)n_occ * xcd) >> 3), hi = (uint32_t)(((uint64_t)n_occ * (xcd + 1u)) >> 3);
	const uint32_t stride = (gridDim.x >> 3) * Q_WAVES;
	uint32_t ci = lo + (blockIdx.x >> 3) * Q_WAVES + w;
	PoolState ps = { 0u, 0u, 0u, 0u, 0u };
	uint32_t wave_hits = 0;

	// software pipeline: occ entry two cells ahead, 27 lookups one cell ahead
	uint2 oc = ci < hi ? a.occ_i[ci] : make_uint2(0u, 0u);
	uint2 oc_n = (ci + stride) < hi ? a.occ_i[ci + stride] : make_uint2(0u, 0u);
	uint32_t s, e;
	lookup_cell<SPARSE>(a, oc.y, ci < hi, lane, s, e);
	uint2 qrange = ci < hi ? own_range<SPARSE>(a, oc.y) : make_uint2(0u, 0u);

	while (ci < hi) {
		const uint32_t ci_n = ci + stride, ci_nn = ci_n + stride;
		const uint2 oc_nn = ci_nn < hi ? a.occ_i[ci_nn] : make_uint2(0u, 0u);

		// ---- merge the 27 lookups of the CURRENT cell into 9 x-runs (lanes 0,3,..,24)
		// (four ds_bpermute, 24 cycles each by tools/ubench/valu_rate.hip; wave_shl:1 DPP moves in their place measured 3 % SLOWER
		//  on the whole query in three interleaved A/Bs of rotating order, profiles/r3_query_ab_micro.txt)
		const uint32_t s1 = __shfl_down(s, 1, WAVE), e1 = __shfl_down(e, 1, WAVE);
		const uint32_t s2 = __shfl_down(s, 2, WAVE), e2 = __shfl_down(e, 2, WAVE);
		RunRef RR;
		RR.run_start = (e > s) ? s : ((e1 > s1) ? s1 : s2);
		const uint32_t run_end = (e2 > s2) ? e2 : ((e1 > s1) ? e1 : e);
		RR.run_len = ((e > s) || (e1 > s1) || (e2 > s2)) ? run_end - RR.run_start : 0u;   // empty entries may hold any (s,s)
		{
			const Runs R0 = extract_runs(RR.run_start, RR.run_len);
			RR.total = R0.total;
		}
		const uint2 cur_q = qrange;

		// ---- issue the lookups of the NEXT cell now; they complete under this cell's arithmetic
		lookup_cell<SPARSE>(a, oc_n.y, ci_n < hi, lane, s, e);
		qrange = ci_n < hi ? own_range<SPARSE>(a, oc_n.y) : make_uint2(0u, 0u);

		// ---- query points of this cell, 64 at a time
		for (uint32_t qb = cur_q.x; qb < cur_q.y; qb += WAVE) {
			const uint32_t nq_all = (cur_q.y - qb) < (uint32_t)WAVE ? (cur_q.y - qb) : (uint32_t)WAVE;
			float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
			float qr2 = a.r2_fixed;
			uint64_t my_off = 0;
			uint32_t qorig = 0;   // original index of the query (the w component, unless the points carry user ids there)
			if ((uint32_t)lane < nq_all) {
				qv = a.xyzi_i[qb + lane];
				qorig = a.orig_i ? a.orig_i[qb + lane] : __float_as_uint(qv.w);
				if (VARIABLE) qr2 = a.r2_i[qb + lane];
				if (MODE == MODE_FILL) my_off = a.offs_sorted[qb + lane];   // start of the record (its count word)
			}
			// queries that want lists (original index < query_limit): a prefix of the cell, the sort is stable (see fast_query_loop)
			const uint32_t nq = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((uint32_t)lane < nq_all && qorig < a.query_limit));
			if (MODE == MODE_COUNT && (uint32_t)lane >= nq && (uint32_t)lane < nq_all) a.counts[qb + lane] = 0u;   // no record at all
			uint32_t run_cnt = 0;
			if (RR.total <= (uint32_t)Q_SLOTS) {
				// ---- the normal case: all candidates of the cell fit into one register-resident batch
				if (RR.total > 0) {
					process_batch_nc<ARITH, VARIABLE, SYM, SELF, MODE, true>(a, RR, 0u, RR.total, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, wave_hits);
				}
				else if (MODE == MODE_POOL && a.shared_empty != 0u) { my_off = ~0ull; }   // (offsets pre-set to the shared empty record: nothing to write)
				else if (MODE == MODE_POOL) {
					// no candidates at all: every query still gets its (empty) record
					for (uint32_t t = 0; t < nq; t++) {
						bool ok1;
						const uint64_t off = pool_alloc<true>(a, ps, 1u, lane, ok1);
						if ((uint32_t)lane == t) my_off = ok1 ? off : ~0ull;
					}
				}
			}
			else if (MODE != MODE_POOL) {
				for (uint32_t wb = 0; wb < RR.total; wb += Q_SLOTS) {
					process_batch_nc<ARITH, VARIABLE, SYM, SELF, MODE, false>(a, RR, wb, Q_SLOTS, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, wave_hits);
				}
			}
			else {
				// ---- rare: several candidate batches in pool mode.  Count sweep, allocate the records of all nq queries
				//      at once, fill sweep.
				//      (the hits are tallied by the count sweep, so that a pass that cannot write -- pool overflow, dry pass -- still
				//      reports how many neighbours there are)
				uint32_t unused_hits = 0;
				for (uint32_t wb = 0; wb < RR.total; wb += Q_SLOTS) {
					process_batch_nc<ARITH, VARIABLE, SYM, SELF, MODE_COUNT, false>(a, RR, wb, Q_SLOTS, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, wave_hits);
				}
				const uint32_t len = (uint32_t)lane < nq ? run_cnt + 1u : 0u;
				uint32_t inc = len;
				#pragma unroll
				for (int o = 1; o < WAVE; o <<= 1) { const uint32_t tv = __shfl_up(inc, o, WAVE); if (lane >= o) inc += tv; }
				const uint32_t total_len = readlane_u32(inc, WAVE - 1);
				bool okm;
				const uint64_t base = pool_alloc<true>(a, ps, total_len, lane, okm);
				my_off = okm ? base + (inc - len) : ~0ull;
				run_cnt = 0;
				if (okm) {
					for (uint32_t wb = 0; wb < RR.total; wb += Q_SLOTS) {
						process_batch_nc<ARITH, VARIABLE, SYM, SELF, MODE_FILL, false>(a, RR, wb, Q_SLOTS, lane, qv, qr2, qb, nq, my_off, run_cnt, ps, unused_hits);
					}
				}
			}
			if ((uint32_t)lane < nq) {
				if (MODE == MODE_COUNT) {
					a.counts[qb + lane] = run_cnt + 1u;
				}
				else if (MODE == MODE_FILL || my_off != ~0ull) {
					a.records[my_off] = (int)run_cnt;
					a.offs_by_orig[qorig] = my_off;
				}
			}
		}
		ci = ci_n; oc = oc_n; oc_n = oc_nn;
	}
	if (MODE == MODE_POOL) pool_wave_done<true>(a, ps, wave_hits, lane);   // (pool mode: this kernel is the general tier)
}

// =====================================================================================================
// pool-mode FAST kernel: the steady-state hot path.
//   * dynamic scheduling: every XCD owns one contiguous eighth of the occupied-cell list and hands it out cell by cell
//     through its own atomic counter, so any number of resident waves stays busy to the end (no static-partition tail,
//     clustered clouds balance themselves) and the waves of an XCD always work on neighbouring cells;
//   * handles only "simple" cells (all candidates in one register batch, at most 64 query points); the others are
//     appended to a worklist that the general kernel above processes afterwards -- this keeps registers low.
// =====================================================================================================

// Instruction budget notes (tools/ubench/issue_model.hip, MI355X): a scalar instruction costs ~4.3 SIMD cycles -- as much
// as a v_cmp or a packed-fp32 op, more than a plain VALU op (2.6) -- and overlaps only partly with VALU work of other
// waves.  The fast path below therefore minimises the TOTAL instruction count per query point:
//   * the allocator state lives in three scalars (record pointer, ints left, slab-inside-pool flag): one 32-bit compare
//     and a pointer bump per query;
//   * no 64-bit compares or offset/pointer conversions inside the loop; the per-lane record pointer is turned into an
//     offset once per cell.
// The query loop of one simple cell: NC register-resident candidate chunks (packed pairs) against the nq query points held one
// per lane in qv / qr2.  Tests, record allocation, emission.
#ifndef TNSX_NT_OFFS
#define TNSX_NT_OFFS 0
#endif
#ifndef TNSX_BLOCK_STORE_HINT
#define TNSX_BLOCK_STORE_HINT " nt"
#endif
template <int NC> struct StageSize { static constexpr uint32_t ints = NC > 8 ? 2048u : 1536u; };   // ints of a wave's staging area: > 2 x the longest record of the tier
// ---------------------------------------------------------------------------------------------------------------------
// Whole-cell staging (round 3).  The records of the queries of one cell are consecutive in the pool, so the WHOLE BLOCK is built in
// the wave's LDS staging area -- [count, ids...] [count, ids...] ... exactly as it will lie in memory -- and leaves at the end of the
// cell: ONE allocation per cell (not per query), full-wave coalesced stores of 64 consecutive ints bounded by the buffer's
// NUM_RECORDS (no compare, no exec moves), one store of the offsets.  Per query nothing is left but the tests, the compaction
// into LDS, one count word (lane 0, through M0: ds_write_addtid needs no address register) and the bump of the staging position.
// Against the per-query version of round 2 (record read back and stored while the next query is tested): no per-query store
// pipeline, no slab test, no flags in vector registers -- see profiles/r3_query_ab.txt.
// ---------------------------------------------------------------------------------------------------------------------
#define TNSX_LDS_PIECE4(K)                  \
	"s_mov_b64 exec, %[m" #K "]\n\t"       \
	"ds_write_b32 %[addr], %[v" #K "] offset:4\n\t" \
	"v_add_u32 %[addr], 4, %[addr]\n\t"
template <int N>
__device__ __forceinline__ void stage_chunks4(uint32_t& addr, const uint64_t* m, const uint32_t* v)
{
	if (N == 1) {
		asm volatile(TNSX_LDS_PIECE4(0) "s_mov_b64 exec, -1" : [addr] "+v"(addr) : TNSX_LDS_IN(0, m[0], v[0]) : "memory");
	}
	else if (N == 2) {
		asm volatile(TNSX_LDS_PIECE4(0) TNSX_LDS_PIECE4(1) "s_mov_b64 exec, -1" : [addr] "+v"(addr) : TNSX_LDS_IN(0, m[0], v[0]), TNSX_LDS_IN(1, m[1], v[1]) : "memory");
	}
	else if (N == 3) {
		asm volatile(TNSX_LDS_PIECE4(0) TNSX_LDS_PIECE4(1) TNSX_LDS_PIECE4(2) "s_mov_b64 exec, -1"
		             : [addr] "+v"(addr) : TNSX_LDS_IN(0, m[0], v[0]), TNSX_LDS_IN(1, m[1], v[1]), TNSX_LDS_IN(2, m[2], v[2]) : "memory");
	}
	else {
		asm volatile(TNSX_LDS_PIECE4(0) TNSX_LDS_PIECE4(1) TNSX_LDS_PIECE4(2) TNSX_LDS_PIECE4(3) "s_mov_b64 exec, -1"
		             : [addr] "+v"(addr) : TNSX_LDS_IN(0, m[0], v[0]), TNSX_LDS_IN(1, m[1], v[1]), TNSX_LDS_IN(2, m[2], v[2]), TNSX_LDS_IN(3, m[3], v[3]) : "memory");
	}
}
template <int NC>
__device__ __forceinline__ void stage_all4(uint32_t& addr, const uint64_t (&m)[NC], const uint32_t* v)
{
	#pragma unroll
	for (int g = 0; g < NC; g += 4) {
		if (NC - g >= 4) stage_chunks4<4>(addr, m + g, v + g);
		else if (NC - g == 3) stage_chunks4<3>(addr, m + g, v + g);
		else if (NC - g == 2) stage_chunks4<2>(addr, m + g, v + g);
		else stage_chunks4<1>(addr, m + g, v + g);
	}
}

template <int ARITH, bool VARIABLE, bool SYM, bool SELF, int NC>
__device__ __forceinline__ void fast_query_loop(const QueryArgs& a, const RunRef RR, int lane, const uint2 cur_q, PoolState& ps, uint32_t& wave_hits,
                                                const v2f (&cx)[(NC + 1) / 2], const v2f (&cy)[(NC + 1) / 2], const v2f (&cz)[(NC + 1) / 2],
                                                const uint32_t (&cid)[2 * ((NC + 1) / 2)], const float (&cr2)[2 * ((NC + 1) / 2)], const float4 qv,
                                                const float qr2, const uint32_t qidx)
{
	constexpr int NP = (NC + 1) / 2;
	constexpr uint32_t STAGE = StageSize<NC>::ints;
	// Only points with original index < query_limit get lists (tnsx_set_query_count: the tail of a set can be candidates only,
	// e.g. the ghost points of a slab).  The cell sort is stable, so inside a cell these queries come first: a prefix.
	const uint32_t nq = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((uint32_t)lane < cur_q.y - cur_q.x && qidx < a.query_limit));
	uint32_t left = readfirstlane_u32(ps.left);
	uint32_t ok = readfirstlane_u32(ps.ok);
	uint64_t base = ((uint64_t)readfirstlane_u32(ps.cur_hi) << 32) | readfirstlane_u32(ps.cur_lo);
	uint32_t* const stage = record_stage<(int)STAGE>();
	const uint32_t stage_base = readfirstlane_u32((uint32_t)(uintptr_t)stage);   // (the low half of a generic LDS address is the LDS offset)
	const uint32_t max_len = RR.total + 1u;   // no record of this cell is longer (<= STAGE / 2 by the tiers' limits)

	// Round 6: the bookkeeping of the loop is in LDS BYTE ADDRESSES -- `rec` = where the current record starts, lane 63's final address = where it ends -- and what
	// a record needs beyond its hits (its count word, its position in the block) is parked in two vector registers, one lane per query, and written when the block
	// leaves: per query two scalar instructions (end - rec, end + 4) instead of eight and no LDS store for the count word (C2 query -1.9 %, then another -0.4 % and C3 -3.3 %:
	// profiles/r6_reg_cull.txt; the scalar unit is shared by the four SIMDs of a CU and this kernel issues ~45 scalar instructions per query).
	uint32_t rec = stage_base;   // LDS byte address of the next record
	uint32_t t0 = 0;             // first query of the block
	uint32_t v_rec = 0;          // lane t: LDS byte address of record t
	uint32_t v_len = 0;          // lane t: 4 x the number of neighbours of query t
	uint32_t hits = 0;
	// the block [stage_base, rec) = the records of queries [t0, t1) -> the pool
	auto flush = [&](uint32_t t1) {
		const uint32_t block = (rec - stage_base) >> 2;
		const uint32_t v_pos = (v_rec - stage_base) >> 2;   // lane t: where record t starts inside its block (ints)
		uint64_t dst = base;        // where this block goes
		uint32_t dst_ok = ok;
		bool from_slab = true;
		if (block > left) {
			// rare: one atomic on the cursor of this XCD's region.  A block of more than half a slab gets an allocation of exactly its
			// size and the wave keeps what is left of its slab for the blocks to come (blocks are whole cells now, hundreds of ints: throwing
			// the rest of a slab away for each of them left the pools of small sets half empty); a smaller block opens a new slab.
			const uint32_t slab = NC > 8 ? a.pool_slab_heavy : a.pool_slab;
			const bool exact = 2u * block > slab;
			if (!exact) pool_waste(ps, left);
			const uint32_t sz = exact ? block : slab;
			const unsigned long long first = pool_take_slab(a.pool_cursor, a.pool_regions, sz, NC > 8 ? 1u : 0u);   // (more than 8 chunks: the fat tier)
			const uint64_t got = ((uint64_t)readfirstlane_u32((uint32_t)(first >> 32)) << 32) | readfirstlane_u32((uint32_t)first);
			const uint32_t got_ok = got != POOL_NONE ? 1u : 0u;
			if (exact) { dst = got_ok ? got : 0; dst_ok = got_ok; from_slab = false; }
			else { base = got_ok ? got : 0; ok = got_ok; left = sz; dst = base; dst_ok = ok; }
		}
		if ((uint32_t)lane >= t0 && (uint32_t)lane < t1) stage[v_pos] = v_len >> 2;   // the count words of the block's records (LDS operations of a wave execute in order)
		asm volatile("" ::: "memory");
		if (dst_ok != 0u) {
			v4i rsrc = record_rsrc(a.records + dst);
			rsrc.z = (int)block;   // NUM_RECORDS: the hardware drops the lanes of the last store that lie beyond the block
			// four reads in flight per round trip to the LDS (STAGE is a multiple of 256: the reads stay inside the wave's area)
			for (uint32_t f = 0; f < block; f += 4u * (uint32_t)WAVE) {
				const uint32_t i = f + (uint32_t)lane;
				const uint32_t v0 = stage[i], v1 = stage[i + 64u], v2 = stage[i + 128u], v3 = stage[i + 192u];
				// (every store is bounded by its own index: the instruction offset takes no part in the hardware's range check)
				asm volatile("buffer_store_dword %[v0], %[i0], %[rsrc], 0 idxen" TNSX_BLOCK_STORE_HINT "\n\t"
				             "buffer_store_dword %[v1], %[i1], %[rsrc], 0 idxen" TNSX_BLOCK_STORE_HINT "\n\t"
				             "buffer_store_dword %[v2], %[i2], %[rsrc], 0 idxen" TNSX_BLOCK_STORE_HINT "\n\t"
				             "buffer_store_dword %[v3], %[i3], %[rsrc], 0 idxen" TNSX_BLOCK_STORE_HINT
				             : : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [i0] "v"(i), [i1] "v"(i + 64u), [i2] "v"(i + 128u), [i3] "v"(i + 192u),
				                 [rsrc] "s"(rsrc) : "memory");
			}
			if ((uint32_t)lane >= t0 && (uint32_t)lane < t1) {
#if TNSX_NT_OFFS
				__builtin_nontemporal_store((unsigned long)(dst + v_pos), reinterpret_cast<unsigned long*>(a.offs_by_orig) + qidx);
#else
				a.offs_by_orig[qidx] = dst + v_pos;
#endif
			}
		}
		hits += block - (t1 - t0);
		if (from_slab) { base += block; left -= block; }
		t0 = t1;
		rec = stage_base;
	};
	const uint32_t rec_limit = stage_base + 4u * (STAGE - max_len);   // a record that starts behind this address may not fit the staging area

	for (uint32_t t = 0; t < nq; t++) {
		if (rec > rec_limit) flush(t);   // (cells with many query points: the block leaves in pieces)
		// (three v_readlane, 4.3 cycles each.  The point through a scalar load instead: -0.5 %; through the LDS -- parked once per cell,
		//  one broadcast ds_read per query -- +8 %: the 4 KB per workgroup it needs take the staging area to the LDS limit of five
		//  workgroups per CU.  profiles/r3_query_ab_micro.txt)
		const float qx = readlane_f32(qv.x, (int)t), qy = readlane_f32(qv.y, (int)t), qz = readlane_f32(qv.z, (int)t);
		const float r2q = VARIABLE ? readlane_f32(qr2, (int)t) : a.r2_fixed;
		uint64_t m[NC];
		#pragma unroll
		for (int h = 0; h < NP; h++) {
			const v2f d2 = dist_sq2<ARITH>(qx, qy, qz, cx[h], cy[h], cz[h]);
			#pragma unroll
			for (int u = 0; u < 2; u++) {
				const int k = 2 * h + u;
				if (k < NC) {
					m[k] = __builtin_amdgcn_ballot_w64(d2[u] <= (SYM ? max_raw(r2q, cr2[k]) : r2q));   // (SYM: see process_batch)
				}
			}
		}
		if (SELF) {
			// the query is always a hit of itself (d2 == 0) and sits at slot t (own-cell-first slot order)
			asm("s_bitset0_b64 %0, %1" : "+s"(m[0]) : "s"(t));
		}
		// hits -> the staging area, lane-major behind the count word: lane l writes at rec + 4 + 4 * (hits of all chunks in lower lanes)
		uint32_t addr;
		{
			uint32_t P = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[0] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[0], 0u));
			#pragma unroll
			for (int k = 1; k < NC; k++) P = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[k], P));
			// (spelled out: left to itself the compiler adds spos and the staging base in two vector instructions)
			asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(addr) : "v"(P), "s"(rec));
		}
		stage_all4<NC>(addr, m, cid);
		const uint32_t end = readlane_u32(addr, WAVE - 1);   // rec + 4 * hits
		asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0" : "+v"(v_rec), "+v"(v_len) : "s"(rec), "s"(end - rec), "s"(t));
		rec = end + 4u;
	}
	if (rec != stage_base) flush(nq);
	wave_hits += hits;
	ps.cur_lo = (uint32_t)base; ps.cur_hi = (uint32_t)(base >> 32);
	ps.left = left;
	ps.ok = ok;
}

template <int ARITH, bool VARIABLE, bool SYM, bool SELF, int NC>
__device__ __forceinline__ void fast_cell(const QueryArgs& a, const RunRef RR, int lane, const uint2 cur_q, PoolState& ps, uint32_t& wave_hits)
{
	constexpr int NP = (NC + 1) / 2;
	constexpr bool OWN_FIRST = SELF;
	const uint32_t nq = cur_q.y - cur_q.x;
	const uint32_t* const tbl = record_stage<(int)StageSize<NC>::ints>();
	struct { uint32_t total, d0; } R;
	R.total = RR.total;
	{
		uint32_t p1_unused;
		if (!TNSX_DEAL_DIRECT || NC > 8 || !deal_runs_direct<OWN_FIRST>(readfirstlane_u32((uint32_t)(uintptr_t)tbl), RR, cur_q.x, (uint32_t)lane, p1_unused, R.d0)) {
			const Runs RF = extract_runs_t<OWN_FIRST>(RR.run_start, RR.run_len, cur_q.x);
			deal_table<OWN_FIRST>(readfirstlane_u32((uint32_t)(uintptr_t)tbl), RF, (uint32_t)lane);
			R.d0 = RF.d0;
		}
	}
	// ---- candidates -> registers (branch-free, see process_batch)
	v2f cx[NP], cy[NP], cz[NP];
	uint32_t cid[2 * NP];
	float cr2[2 * NP];
	float4 craw[2 * NP];
	float r2raw[2 * NP];
	#pragma unroll
	for (int k = 0; k < 2 * NP; k++) {
		if (k < NC) {
			const uint32_t slot = (uint32_t)(k * WAVE + lane);
			const uint32_t src = (k < NC - 1 || slot < R.total) ? tbl[slot] : R.d0;
			craw[k] = a.xyzi_j[src];
			if (SYM) r2raw[k] = a.r2_j[src];
		}
	}
	// the cell's query points, one per lane (clamped, branch-free load)
	const uint32_t qsrc = cur_q.x + ((uint32_t)lane < nq ? (uint32_t)lane : 0u);
	const float4 qv = a.xyzi_i[qsrc];
	const uint32_t qorig = a.orig_i ? a.orig_i[qsrc] : __float_as_uint(qv.w);
	float qr2 = a.r2_fixed;
	if (VARIABLE) qr2 = a.r2_i[qsrc];
	#pragma unroll
	for (int k = 0; k < 2 * NP; k++) {
		float4 c = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, __uint_as_float(0xffffffffu));
		float r2c = -1.0f;
		if (k < NC) {
			c = craw[k];
			if (SYM) r2c = r2raw[k];
			if (k == NC - 1) {   // the chunk count is exact: only the last chunk can be partial
				const bool valid = (uint32_t)(k * WAVE + lane) < R.total;
				c.x = valid ? c.x : FLT_MAX; c.y = valid ? c.y : FLT_MAX; c.z = valid ? c.z : FLT_MAX;
				c.w = valid ? c.w : __uint_as_float(0xffffffffu);
				if (SYM) r2c = valid ? r2c : -1.0f;
			}
		}
		cx[k >> 1][k & 1] = c.x; cy[k >> 1][k & 1] = c.y; cz[k >> 1][k & 1] = c.z;
		cid[k] = __float_as_uint(c.w);
		cr2[k] = r2c;
	}

	fast_query_loop<ARITH, VARIABLE, SYM, SELF, NC>(a, RR, lane, cur_q, ps, wave_hits, cx, cy, cz, cid, cr2, qv, qr2, qorig);
}

// ---------------------------------------------------------------------------------------------------------------------
// Bounding-box cull (first tier, cells with 513..1024 candidates: dense fluid, or a cell edge r_max well above most radii).
// Every candidate is tested once against the bounding box of the cell's query points -- a lower bound of the squared distance
// in the predicate's own arithmetic.  About a third of the candidates goes, and what is left nearly always fits the eight
// chunks of the query loop, so these cells stay in the first tier (5 waves per SIMD) instead of the 16-chunk second
// tier (3 waves).  The candidates are looked at eight chunks at a time, so no more registers are needed than the loop has; only
// the SLOT NUMBERS of the survivors are staged (2 bytes each in LDS) and the survivors are loaded a second time, from the L2.
// More than 512 survivors: nothing has been written, the cell goes to the second tier.
// (For cells that fit the loop anyway the cull was measured to pay nothing: 6.4 -> 4.3 chunks per query, but the staging
// per cell costs what the shorter loop saves.)
//
// Exactness: every fp32 op of the predicate is monotone.  With b = max(fl(lo - c), fl(c - hi), 0) per axis, |fl(q - c)| >= b for
// every query coordinate q in [lo, hi]; squares, sums and fmas preserve that order, so d2(q, c) >= d2lb(c) for all queries of the
// cell, and d2lb > (largest radius^2 that can accept the pair) means that the predicate itself rejects the pair.
// The centre row (slots < R.p1: it holds the cell's own points) is kept as it is, so that the self-exclusion slot stays valid.
// ---------------------------------------------------------------------------------------------------------------------
template <int ARITH, bool SYM, bool OWN_FIRST>
__device__ __forceinline__ uint32_t cull_round(const QueryArgs& a, const Runs R, int lane, uint32_t base, uint32_t kept, float lox, float loy, float loz,
                                               float hix, float hiy, float hiz, float r2q_max, uint16_t* __restrict__ lds_slots, const uint32_t* __restrict__ tbl,
                                               uint32_t slot_cap)
{
	const uint32_t nc = (R.total - base + WAVE - 1) / WAVE;   // chunks of this round that hold candidates (the rest is skipped)
	// (only the coordinates: the cull never looks at a candidate's index, and the eight registers a dwordx4 load would keep for it are what spilled around
	//  this path in the instantiations with per-point radii)
	struct Xyz { float x, y, z; };
	Xyz craw[Q_MAXPAIRS * 2];
	float r2raw[Q_MAXPAIRS * 2];
	#pragma unroll
	for (int k = 0; k < Q_MAXPAIRS * 2; k++) {
		if ((uint32_t)k < nc) {
			const uint32_t slot = base + (uint32_t)(k * WAVE + lane);
			const uint32_t src = slot < R.total ? tbl[slot] : R.d0;
			craw[k] = *reinterpret_cast<const Xyz*>(a.xyzi_j + src);
			if (SYM) r2raw[k] = a.r2_j[src];
		}
	}
	#pragma unroll
	for (int k = 0; k < Q_MAXPAIRS * 2; k++) {
		if ((uint32_t)k < nc) {
			const uint32_t slot = base + (uint32_t)(k * WAVE + lane);
			const Xyz c = craw[k];
			const float bx = fmaxf(fmaxf(__fsub_rn(lox, c.x), __fsub_rn(c.x, hix)), 0.0f);
			const float by = fmaxf(fmaxf(__fsub_rn(loy, c.y), __fsub_rn(c.y, hiy)), 0.0f);
			const float bz = fmaxf(fmaxf(__fsub_rn(loz, c.z), __fsub_rn(c.z, hiz)), 0.0f);
			float d2lb;
			if (ARITH == 0) d2lb = __fadd_rn(__fadd_rn(__fmul_rn(bx, bx), __fmul_rn(by, by)), __fmul_rn(bz, bz));
			else d2lb = __fmaf_rn(bz, bz, __fmaf_rn(bx, bx, __fmul_rn(by, by)));
			const float lim = SYM ? fmaxf(r2q_max, r2raw[k]) : r2q_max;
			const bool keep = slot < R.total && (slot < R.p1 || d2lb <= lim);
			const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
			const uint32_t pos = kept + mbcnt64(m);
			if (keep && pos < slot_cap) lds_slots[pos] = (uint16_t)slot;
			kept += (uint32_t)__popcll(m);
		}
	}
	return kept;
}

// the query loop on the `kept` surviving candidates whose slot numbers are in lds_slots
template <int ARITH, bool VARIABLE, bool SYM, bool SELF, int NC>
__device__ __forceinline__ void fast_cell_from_slots(const QueryArgs& a, const RunRef RR, const Runs R, int lane, const uint2 cur_q, PoolState& ps,
                                                     uint32_t& wave_hits, const uint16_t* __restrict__ lds_slots, uint32_t kept, const float4 qv, const float qr2,
                                                     const uint32_t qorig)
{
	constexpr int NP = (NC + 1) / 2;
	v2f cx[NP], cy[NP], cz[NP];
	uint32_t cid[2 * NP];
	float cr2[2 * NP];
	float4 craw[2 * NP];
	float r2raw[2 * NP];
	const uint32_t* const tbl = record_stage<(int)StageSize<NC>::ints>();   // (the table fast_cell_culled built; NC <= 8: the first tier's area)
	(void)tbl;
	#pragma unroll
	for (int k = 0; k < 2 * NP; k++) {
		if (k < NC) {
			const uint32_t i = (uint32_t)(k * WAVE + lane);
			const uint32_t slot = lds_slots[i];                     // (slots past `kept` hold stale numbers: clamped below)
			const uint32_t src = (k < NC - 1 || i < kept) ? tbl[slot < R.total ? slot : 0u] : R.d0;
			craw[k] = a.xyzi_j[src];
			if (SYM) r2raw[k] = a.r2_j[src];
		}
	}
	#pragma unroll
	for (int k = 0; k < 2 * NP; k++) {
		float4 c = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, __uint_as_float(0xffffffffu));
		float r2c = -1.0f;
		if (k < NC) {
			c = craw[k];
			if (SYM) r2c = r2raw[k];
			if (k == NC - 1) {   // the chunk count is exact: only the last chunk can be partial
				const bool valid = (uint32_t)(k * WAVE + lane) < kept;
				c.x = valid ? c.x : FLT_MAX; c.y = valid ? c.y : FLT_MAX; c.z = valid ? c.z : FLT_MAX;
				c.w = valid ? c.w : __uint_as_float(0xffffffffu);
				if (SYM) r2c = valid ? r2c : -1.0f;
			}
		}
		cx[k >> 1][k & 1] = c.x; cy[k >> 1][k & 1] = c.y; cz[k >> 1][k & 1] = c.z;
		cid[k] = __float_as_uint(c.w);
		cr2[k] = r2c;
	}
	fast_query_loop<ARITH, VARIABLE, SYM, SELF, NC>(a, RR, lane, cur_q, ps, wave_hits, cx, cy, cz, cid, cr2, qv, qr2, qorig);
}

// -> false: more candidates survive than the tier's loop holds (first tier: 512, fat tier: 1024); nothing has been written.
// Round 3: the fat tier culls too.  Its cells are the ones of which more than 512 candidates survived the first tier's cull; the
// cull's result is not kept (the survivors' slot numbers live in the first-tier wave's LDS), so the fat wave repeats it -- and then
// runs its query loop on the ~2/3 of the candidates that survive instead of all of them (C4: the dense column of the dam break).
template <int ARITH, bool VARIABLE, bool SYM, bool SELF, bool FAT>
__device__ __forceinline__ bool fast_cell_culled(const QueryArgs& a, const RunRef RR, int lane, const uint2 cur_q, PoolState& ps, uint32_t& wave_hits,
                                                 uint16_t* __restrict__ lds_slots)
{
	constexpr uint32_t SLOT_CAP = FAT ? 2u * (uint32_t)Q_SLOTS : (uint32_t)Q_SLOTS;
	const uint32_t nq = cur_q.y - cur_q.x;
	constexpr bool OWN_FIRST = SELF;
	const Runs R = extract_runs_t<OWN_FIRST>(RR.run_start, RR.run_len, cur_q.x);
	// the cell's query points, one per lane (clamped, branch-free load: the lanes beyond nq repeat the first point)
	const uint32_t qsrc = cur_q.x + ((uint32_t)lane < nq ? (uint32_t)lane : 0u);
	const float4 qv = a.xyzi_i[qsrc];
	const uint32_t qorig = a.orig_i ? a.orig_i[qsrc] : __float_as_uint(qv.w);
	float qr2 = a.r2_fixed;
	if (VARIABLE) qr2 = a.r2_i[qsrc];
	float lox = qv.x, loy = qv.y, loz = qv.z, hix = qv.x, hiy = qv.y, hiz = qv.z;
	wave_bbox(lox, loy, loz, hix, hiy, hiz);
	const float r2q_max = VARIABLE ? wave_max_dpp(qr2) : a.r2_fixed;

	const uint32_t* const tbl = record_stage<(int)StageSize<(FAT ? 16 : 8)>::ints>();
	deal_table<OWN_FIRST>(readfirstlane_u32((uint32_t)(uintptr_t)tbl), R, (uint32_t)lane);   // up to 1024 slots: inside the staging area
	uint32_t kept = 0;
	for (uint32_t base = 0; base < R.total; base += (uint32_t)Q_SLOTS)
		kept = readfirstlane_u32(cull_round<ARITH, SYM, OWN_FIRST>(a, R, lane, base, kept, lox, loy, loz, hix, hiy, hiz, r2q_max, lds_slots, tbl, SLOT_CAP));
	if (kept > SLOT_CAP) return false;
	if (FAT && kept <= (uint32_t)Q_SLOTS) return false;   // (cannot happen: the first tier would have taken the cell; the plain fat path is correct anyway)
	wave_lds_fence();
#define TNSX_FROM_SLOTS(N) fast_cell_from_slots<ARITH, VARIABLE, SYM, SELF, N>(a, RR, R, lane, cur_q, ps, wave_hits, lds_slots, kept, qv, qr2, qorig)
	if (!FAT) {
		switch ((kept + WAVE - 1) / WAVE) {
		case 0:
		case 1: TNSX_FROM_SLOTS(1); break;
		case 2: TNSX_FROM_SLOTS(2); break;
		case 3: TNSX_FROM_SLOTS(3); break;
		case 4: TNSX_FROM_SLOTS(4); break;
		case 5: TNSX_FROM_SLOTS(5); break;
		case 6: TNSX_FROM_SLOTS(6); break;
		case 7: TNSX_FROM_SLOTS(7); break;
		default: TNSX_FROM_SLOTS(8); break;
		}
	}
	else {
		switch ((kept + WAVE - 1) / WAVE) {
		case 9: TNSX_FROM_SLOTS(9); break;
		case 10: TNSX_FROM_SLOTS(10); break;
		case 11: TNSX_FROM_SLOTS(11); break;
		case 12: TNSX_FROM_SLOTS(12); break;
		case 13: TNSX_FROM_SLOTS(13); break;
		case 14: TNSX_FROM_SLOTS(14); break;
		case 15: TNSX_FROM_SLOTS(15); break;
		default: TNSX_FROM_SLOTS(16); break;
		}
	}
#undef TNSX_FROM_SLOTS
	wave_lds_fence();   // the next culled cell of this wave overwrites the staging buffer
	return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: bounding-box cull IN REGISTERS for cells that fit the first tier's loop anyway (5..7 chunks: C2, C3, C5), survivors compacted through the
// wave's LDS staging area.  The cull above (fast_cell_culled) was measured to LOSE on such cells (C2 1.58 -> 1.86 ms): it loads the coordinates, stages the
// survivors' SLOT NUMBERS and loads the survivors a second time from global memory through two dependent LDS reads -- a second memory round trip per cell.
// Here the candidates are loaded ONCE, as always; the bound is evaluated on the registers; the surviving float4s are written to the staging area in slot
// order (ds_write_b128; the area holds the deal table before and the cell's records after, it is empty in between) and read back as 4.2 chunks instead of
// 6.4 for the query loop.  Own cell first: the centre run [own | x + 1] is kept whole and the compaction preserves the slot order, so query t still is
// candidate slot t.  Fixed radius and per-point radii without symmetry (the symmetric predicate needs r_j^2 beside the point: five words per survivor).
// Exactness: b = c - med3(c, lo, hi) is 0 inside [lo, hi], fl(c - lo) below, fl(c - hi) above; |fl(q - c)| >= |b| for every query coordinate q in [lo, hi]
// (rounding is monotone), squares, sums and fmas preserve the order, so d2(q, c) >= d2lb(c) in the predicate's own arithmetic for all queries of the cell.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef TNSX_REG_CULL
#define TNSX_REG_CULL 1
#endif
#ifndef TNSX_REG_CULL_FROM
#define TNSX_REG_CULL_FROM 4   // chunks: below, the cull's per-cell cost exceeds what the shorter loop saves (profiles/r6_reg_cull.txt)
#endif
#ifndef TNSX_REG_CULL_SYM
#define TNSX_REG_CULL_SYM 0    // 1: also for the symmetric predicate of per-point radii (r_j^2 travels through the staging area beside the point).  Measured at C4: +2.3 % (five
                               // words per survivor, 16 bytes of scratch in those instantiations): off (profiles/r6_reg_cull.txt)
#endif
// lanes [0, min(n, 64)) as a scalar mask (n wave-uniform)
__device__ __forceinline__ uint64_t low_lanes(uint32_t n)
{
	uint64_t m;
	const uint32_t ns = readfirstlane_u32(n);   // (an "s" operand the compiler holds in a vector register is not moved for us)
	// ("=&s": the mask is written before n is read for the last time -- without the early-clobber mark the two may share a register, and did in one build of round 6)
	asm("s_bfm_b64 %0, %1, 0\n\ts_cmp_gt_u32 %1, 63\n\ts_cselect_b64 %0, -1, %0" : "=&s"(m) : "s"(ns) : "scc");
	return m;
}
// survivors the staging area holds: float4 per survivor, + its r_j^2 behind the float4s when the predicate is symmetric
template <bool SYM> struct RegCull {
	static constexpr uint32_t cap = SYM ? StageSize<8>::ints / 5u : StageSize<8>::ints / 4u;   // 307 / 384
	static constexpr uint32_t r2_word = cap * 4u;                                              // first word of the r^2 column
};
// squared lower bound of the distance of candidate (cx, cy, cz) to the box [lo, hi], in the predicate's arithmetic.  Spelled out: left to itself the compiler pairs
// the three axes of two chunks into packed instructions and spends as many v_mov on the pairing as the packing saves.
template <int ARITH>
__device__ __forceinline__ float box_d2lb(float cx, float cy, float cz, float lox, float loy, float loz, float hxv, float hyv, float hzv)
{
	float t0, t1, t2;
	if (ARITH == 0) {
		asm("v_med3_f32 %0, %3, %6, %9\n\tv_med3_f32 %1, %4, %7, %10\n\tv_med3_f32 %2, %5, %8, %11\n\t"
		    "v_sub_f32 %0, %3, %0\n\tv_sub_f32 %1, %4, %1\n\tv_sub_f32 %2, %5, %2\n\t"
		    "v_mul_f32 %0, %0, %0\n\tv_mul_f32 %1, %1, %1\n\tv_mul_f32 %2, %2, %2\n\t"
		    "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2"
		    : "=&v"(t0), "=&v"(t1), "=&v"(t2) : "v"(cx), "v"(cy), "v"(cz), "s"(lox), "s"(loy), "s"(loz), "v"(hxv), "v"(hyv), "v"(hzv));
	}
	else {
		asm("v_med3_f32 %0, %3, %6, %9\n\tv_med3_f32 %1, %4, %7, %10\n\tv_med3_f32 %2, %5, %8, %11\n\t"
		    "v_sub_f32 %0, %3, %0\n\tv_sub_f32 %1, %4, %1\n\tv_sub_f32 %2, %5, %2\n\t"
		    "v_mul_f32 %1, %1, %1\n\tv_fmac_f32 %1, %0, %0\n\tv_fmac_f32 %1, %2, %2\n\tv_mov_b32 %0, %1"
		    : "=&v"(t0), "=&v"(t1), "=&v"(t2) : "v"(cx), "v"(cy), "v"(cz), "s"(lox), "s"(loy), "s"(loz), "v"(hxv), "v"(hyv), "v"(hzv));
	}
	return t0;
}
template <int ARITH, bool VARIABLE, bool SYM, bool SELF, int NC>
__device__ __forceinline__ uint32_t cull_cell_to_stage(const QueryArgs& a, const RunRef RR, int lane, const uint2 cur_q, float4& qv, float& qr2, uint32_t& qorig)
{
	constexpr bool OWN_FIRST = SELF;
	constexpr uint32_t CAP = RegCull<SYM>::cap;
	const uint32_t nq = cur_q.y - cur_q.x;
	uint32_t* const tbl = record_stage<(int)StageSize<8>::ints>();
	struct { uint32_t total, p1, d0; } R;
	R.total = RR.total;
	if (!TNSX_DEAL_DIRECT || !deal_runs_direct<OWN_FIRST>(readfirstlane_u32((uint32_t)(uintptr_t)tbl), RR, cur_q.x, (uint32_t)lane, R.p1, R.d0)) {
		const Runs RF = extract_runs_t<OWN_FIRST>(RR.run_start, RR.run_len, cur_q.x);
		deal_table<OWN_FIRST>(readfirstlane_u32((uint32_t)(uintptr_t)tbl), RF, (uint32_t)lane);
		R.p1 = RF.p1; R.d0 = RF.d0;
	}
	float4 craw[NC];
	float r2raw[NC];
	#pragma unroll
	for (int k = 0; k < NC; k++) {
		const uint32_t slot = (uint32_t)(k * WAVE + lane);
		const uint32_t src = (k < NC - 1 || slot < R.total) ? tbl[slot] : R.d0;
		craw[k] = a.xyzi_j[src];
		if (SYM) r2raw[k] = a.r2_j[src];
	}
	const uint32_t qsrc = cur_q.x + ((uint32_t)lane < nq ? (uint32_t)lane : 0u);   // (lanes beyond nq repeat the first point: harmless for the box)
	qv = a.xyzi_i[qsrc];
	qorig = a.orig_i ? a.orig_i[qsrc] : __float_as_uint(qv.w);
	qr2 = a.r2_fixed;
	if (VARIABLE) qr2 = a.r2_i[qsrc];
	float lox = qv.x, loy = qv.y, loz = qv.z, hix = qv.x, hiy = qv.y, hiz = qv.z;
	if (nq <= 16u) wave_bbox16(lox, loy, loz, hix, hiy, hiz);
	else wave_bbox(lox, loy, loz, hix, hiy, hiz);
	const float lim = VARIABLE ? wave_max_dpp(qr2) : a.r2_fixed;
	// (v_med3_f32 may read ONE scalar register: the upper corner of the box lives in three vector registers for the whole cull)
	float hxv = hix, hyv = hiy, hzv = hiz;
	asm volatile("" : "+v"(hxv), "+v"(hyv), "+v"(hzv));
	lox = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lox)));   // ("s" operands of box_d2lb: the compiler does not move one out of a vector register for us)
	loy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(loy)));
	loz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(loz)));
	const uint32_t stage_base = readfirstlane_u32((uint32_t)(uintptr_t)tbl);
	uint32_t kept = 0;
	bool fits = true;
	#pragma unroll
	for (int k = 0; k < NC; k++) {
		const float4 c = craw[k];
		const float d2lb = box_d2lb<ARITH>(c.x, c.y, c.z, lox, loy, loz, hxv, hyv, hzv);
		// survivors of this chunk: the bound, OR the slots of the centre run (kept whole: scalar mask), AND the chunk's valid slots (scalar mask)
		uint64_t m = __builtin_amdgcn_ballot_w64(d2lb <= (SYM ? max_raw(lim, r2raw[k]) : lim));
		const uint32_t first = (uint32_t)(k * WAVE);
		const uint32_t own = R.p1 > first ? R.p1 - first : 0u;
		if (k <= 1) m |= low_lanes(own);            // (a centre run of more than 128 slots would be a cell of the heavy tiers)
		else if (own != 0u) { fits = false; break; }
		if (k == NC - 1) m &= low_lanes(R.total - first);
		const uint32_t n = (uint32_t)__popcll(m);
		if (kept + n > CAP) { fits = false; break; }   // (rare: the caller takes the plain path)
		// lane l of the mask -> slot kept + (set bits below l); one ds_write_b128 (and one ds_write_b32 for r_j^2) under the mask
		uint32_t pos = mbcnt64(m);
		const v4f cv = { c.x, c.y, c.z, c.w };
		if (SYM) {
			uint32_t ad2;
			asm volatile("v_lshl_add_u32 %[a2], %[ad], 2, %[base2]\n\t"
			             "v_lshl_add_u32 %[ad], %[ad], 4, %[base]\n\t"
			             "s_mov_b64 exec, %[m]\n\t"
			             "ds_write_b128 %[ad], %[v]\n\t"
			             "ds_write_b32 %[a2], %[r]\n\t"
			             "s_mov_b64 exec, -1"
			             : [ad] "+v"(pos), [a2] "=&v"(ad2)
			             : [base] "s"(readfirstlane_u32(stage_base + (kept << 4))), [base2] "s"(readfirstlane_u32(stage_base + 4u * RegCull<SYM>::r2_word + (kept << 2))), [m] "s"(m),
			               [v] "v"(cv), [r] "v"(r2raw[k]) : "memory");
		}
		else {
			asm volatile("v_lshl_add_u32 %[ad], %[ad], 4, %[base]\n\t"
			             "s_mov_b64 exec, %[m]\n\t"
			             "ds_write_b128 %[ad], %[v]\n\t"
			             "s_mov_b64 exec, -1"
			             : [ad] "+v"(pos) : [base] "s"(readfirstlane_u32(stage_base + (kept << 4))), [m] "s"(m), [v] "v"(cv) : "memory");
		}
		kept += n;
	}
	if (!fits) return CAP + 1u;
	return kept;
}
// the query loop on the `kept` survivors in the staging area
template <int ARITH, bool VARIABLE, bool SYM, bool SELF, int NC>
__device__ __forceinline__ void fast_cell_from_stage(const QueryArgs& a, const RunRef RR, int lane, const uint2 cur_q, PoolState& ps, uint32_t& wave_hits, uint32_t kept,
                                                     const float4 qv, const float qr2, const uint32_t qorig)
{
	constexpr int NP = (NC + 1) / 2;
	const uint32_t* const stw = record_stage<(int)StageSize<8>::ints>();
	const float4* const st = reinterpret_cast<const float4*>(stw);
	const float* const st_r2 = reinterpret_cast<const float*>(stw + RegCull<SYM>::r2_word);
	v2f cx[NP], cy[NP], cz[NP];
	uint32_t cid[2 * NP];
	float cr2[2 * NP];
	float4 craw[NC];
	float r2raw[NC];
	#pragma unroll
	for (int k = 0; k < NC; k++) {   // (slots past `kept` hold stale words of this wave's own area: masked below)
		craw[k] = st[k * WAVE + lane];
		if (SYM) r2raw[k] = st_r2[k * WAVE + lane];
	}
	#pragma unroll
	for (int k = 0; k < 2 * NP; k++) {
		float4 c = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, __uint_as_float(0xffffffffu));
		float r2c = -1.0f;
		if (k < NC) {
			c = craw[k];
			if (SYM) r2c = r2raw[k];
			if (k == NC - 1) {
				const bool valid = (uint32_t)(k * WAVE + lane) < kept;
				c.x = valid ? c.x : FLT_MAX; c.y = valid ? c.y : FLT_MAX; c.z = valid ? c.z : FLT_MAX;
				c.w = valid ? c.w : __uint_as_float(0xffffffffu);
				if (SYM) r2c = valid ? r2c : -1.0f;
			}
		}
		cx[k >> 1][k & 1] = c.x; cy[k >> 1][k & 1] = c.y; cz[k >> 1][k & 1] = c.z;
		cid[k] = __float_as_uint(c.w);
		cr2[k] = r2c;
	}
	fast_query_loop<ARITH, VARIABLE, SYM, SELF, NC>(a, RR, lane, cur_q, ps, wave_hits, cx, cy, cz, cid, cr2, qv, qr2, qorig);
}
template <int ARITH, bool VARIABLE, bool SYM, bool SELF>
__device__ __forceinline__ bool fast_cell_reg_culled(const QueryArgs& a, const RunRef RR, int lane, const uint2 cur_q, PoolState& ps, uint32_t& wave_hits, uint32_t nc)
{
	float4 qv; float qr2; uint32_t qorig, kept;
	static_assert(TNSX_REG_CULL_FROM >= 3, "the chunk counts below");
	switch (nc) {   // (the chunk count is exact: only the last chunk of a body may be partial)
	case 3: kept = cull_cell_to_stage<ARITH, VARIABLE, SYM, SELF, 3>(a, RR, lane, cur_q, qv, qr2, qorig); break;
	case 4: kept = cull_cell_to_stage<ARITH, VARIABLE, SYM, SELF, 4>(a, RR, lane, cur_q, qv, qr2, qorig); break;
	case 5: kept = cull_cell_to_stage<ARITH, VARIABLE, SYM, SELF, 5>(a, RR, lane, cur_q, qv, qr2, qorig); break;
	case 6: kept = cull_cell_to_stage<ARITH, VARIABLE, SYM, SELF, 6>(a, RR, lane, cur_q, qv, qr2, qorig); break;
	default: kept = cull_cell_to_stage<ARITH, VARIABLE, SYM, SELF, 7>(a, RR, lane, cur_q, qv, qr2, qorig); break;
	}
	kept = readfirstlane_u32(kept);
	if (kept > RegCull<SYM>::cap) return false;   // (more survivors than the staging area holds: the caller takes the plain path, which loads the candidates again)
#define TNSX_FROM_STAGE(N) fast_cell_from_stage<ARITH, VARIABLE, SYM, SELF, N>(a, RR, lane, cur_q, ps, wave_hits, kept, qv, qr2, qorig)
	switch ((kept + WAVE - 1) / WAVE) {
	case 0:
	case 1: TNSX_FROM_STAGE(1); break;
	case 2: TNSX_FROM_STAGE(2); break;
	case 3: TNSX_FROM_STAGE(3); break;
	case 4: TNSX_FROM_STAGE(4); break;
	case 5: TNSX_FROM_STAGE(5); break;
	default: TNSX_FROM_STAGE(6); break;
	}
#undef TNSX_FROM_STAGE
	return true;
}

template <int ARITH, bool VARIABLE, bool SYM, bool SELF, bool FAT>
__device__ __forceinline__ void fast_cell_nc(const QueryArgs& a, const RunRef RR, int lane, const uint2 cur_q, PoolState& ps, uint32_t& wave_hits)
{
	const uint32_t nc = (RR.total + WAVE - 1) / WAVE;
	if (!FAT) {
		if (TNSX_REG_CULL && (!SYM || TNSX_REG_CULL_SYM) && nc >= (uint32_t)TNSX_REG_CULL_FROM && nc <= 7u) {
			if (fast_cell_reg_culled<ARITH, VARIABLE, SYM, SELF>(a, RR, lane, cur_q, ps, wave_hits, nc)) return;
		}
		switch (nc) {
		case 1: fast_cell<ARITH, VARIABLE, SYM, SELF, 1>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 2: fast_cell<ARITH, VARIABLE, SYM, SELF, 2>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 3: fast_cell<ARITH, VARIABLE, SYM, SELF, 3>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 4: fast_cell<ARITH, VARIABLE, SYM, SELF, 4>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 5: fast_cell<ARITH, VARIABLE, SYM, SELF, 5>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 6: fast_cell<ARITH, VARIABLE, SYM, SELF, 6>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 7: fast_cell<ARITH, VARIABLE, SYM, SELF, 7>(a, RR, lane, cur_q, ps, wave_hits); break;
		default: fast_cell<ARITH, VARIABLE, SYM, SELF, 8>(a, RR, lane, cur_q, ps, wave_hits); break;
		}
	}
	else {
		// 513..1024 candidates of which more than 512 survive the first tier's cull: same single pass, more registers
		switch (nc) {
		case 9: fast_cell<ARITH, VARIABLE, SYM, SELF, 9>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 10: fast_cell<ARITH, VARIABLE, SYM, SELF, 10>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 11: fast_cell<ARITH, VARIABLE, SYM, SELF, 11>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 12: fast_cell<ARITH, VARIABLE, SYM, SELF, 12>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 13: fast_cell<ARITH, VARIABLE, SYM, SELF, 13>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 14: fast_cell<ARITH, VARIABLE, SYM, SELF, 14>(a, RR, lane, cur_q, ps, wave_hits); break;
		case 15: fast_cell<ARITH, VARIABLE, SYM, SELF, 15>(a, RR, lane, cur_q, ps, wave_hits); break;
		default: fast_cell<ARITH, VARIABLE, SYM, SELF, 16>(a, RR, lane, cur_q, ps, wave_hits); break;
		}
	}
}

// Occupancy of the first tier, pinned: left alone the compiler's register budget changes with every edit of the kernel (85 VGPRs /
// 5 waves at one point, a slower schedule than either pinned variant).  5 waves per SIMD (<= 96 VGPRs, no spills) against 6 (80
// VGPRs, a few spilled dwords), measured with tools/ab_libs.py on the final kernel: C2 equal or -4 %, C3 -4 %, C4 -4 %.
// The second tier keeps its 3 waves (it needs ~135 VGPRs for 16 chunks).
#ifndef TNSX_FAST_WAVES_PER_EU
#define TNSX_FAST_WAVES_PER_EU 5
#endif

#ifndef TNSX_FAT_WAVES_PER_EU
#define TNSX_FAT_WAVES_PER_EU 3
#endif
// Round 5: no first-tier instantiation spills any more (tools/kernel_resources.py: scratch 0; the fixed-radius / self kernel of C2 went from 96 to 84 VGPRs) -- the
// lane-derived constants of the cell bodies are recomputed per cell (TNSX_LANE_OPAQUE), the neighbour offsets of the look-ups are one packed register, the cull loads
// coordinates only.  One instantiation is still a register short at five waves: per-point radii + symmetric + two different sets + contracted arithmetic (both sets'
// pointers live in scalar registers, and what does not fit there sits in vector registers); it alone may drop to four waves (<= 128 VGPRs) instead of spilling.
template <int ARITH, bool VARIABLE, bool SYM, bool SELF, bool FAT>
#if TNSX_FAST_WAVES_PER_EU > 0
__attribute__((amdgpu_waves_per_eu(FAT ? TNSX_FAT_WAVES_PER_EU : ((ARITH == 1 && VARIABLE && SYM && !SELF) ? TNSX_FAST_WAVES_PER_EU - 1 : TNSX_FAST_WAVES_PER_EU),
                                   FAT ? TNSX_FAT_WAVES_PER_EU : TNSX_FAST_WAVES_PER_EU)))
#endif
__global__ void __launch_bounds__(Q_THREADS) k_query_pool_fast(const QueryArgs a)
{
	if (a.abort_flag && *a.abort_flag != 0u) return;   // (this attempt is already known to be wrong; its sorted arrays may have holes)
	// FAT = false: cells from the occupied-cell list, 1..8 chunks; the rest -> a.heavy.
	// FAT = true : cells from a.heavy, 9..16 chunks; the rest -> a.heavy2 (general kernel).
	const uint2* __restrict__ cell_list = FAT ? a.heavy : a.occ_i;
	uint2* __restrict__ reject_list = FAT ? a.heavy2 : a.heavy;
	uint32_t* reject_count = FAT ? a.n_heavy2 : a.n_heavy;
	uint32_t* tickets = FAT ? a.tickets2 : a.tickets;
	// staging buffer of the bounding-box cull (slot numbers of the survivors), one slice per wave
	constexpr int SLOT_CAP = FAT ? 2 * Q_SLOTS : Q_SLOTS;
	__shared__ uint16_t s_slots[TNSX_CULL ? Q_WAVES * SLOT_CAP : 2];
	uint16_t* const my_slots = s_slots + (TNSX_CULL ? (threadIdx.x / WAVE) * SLOT_CAP : 0);
	const int lane = lane_id();
	const uint32_t n_occ = FAT ? *a.n_heavy : *a.n_occ_i;
	const uint32_t xcd = blockIdx.x & 7u;
	const uint32_t lo = (uint32_t)(((uint64_t)n_occ * xcd) >> 3), hi = (uint32_t)(((uint64_t)n_occ * (xcd + 1u)) >> 3);
	// more waves than cells (short worklists of the later tiers): the surplus leaves without touching the counter
	if ((blockIdx.x >> 3) * Q_WAVES + threadIdx.x / WAVE >= hi - lo) return;
	PoolState ps = { 0u, 0u, 0u, 0u, 0u };
	uint32_t wave_hits = 0;

	// One cell per ticket (measured: 1.79 ms against 1.99 ms with tickets of 8 cells), fetched through a three-deep software
	// pipeline so that only the candidate loads of a cell are ever waited for:
	//   ticket (atomic) two cells ahead -> occupied-cell entry one cell ahead -> its 27 lookups issued before the current
	//   cell is processed.
	// One counter can hand out ~88 tickets per microsecond (the L2 serialises the atomics of a cache line), which would cap a
	// pass over millions of cheap cells.  Every XCD's share of the cell list is therefore cut into CTRL_SUBRANGES contiguous
	// pieces with a counter each; a wave starts on piece (its index % CTRL_SUBRANGES) and moves on to the next piece when one
	// is used up, until all are.
	const uint32_t span = hi - lo;
	uint32_t sub = readfirstlane_u32(((blockIdx.x >> 3) * Q_WAVES + threadIdx.x / WAVE) % CTRL_SUBRANGES), used_up = 0;
	auto sub_begin = [&](uint32_t k) { return lo + (uint32_t)(((uint64_t)span * k) / CTRL_SUBRANGES); };
	uint32_t cur_lo = sub_begin(sub), cur_hi = sub_begin(sub + 1u);
	auto take = [&]() { uint32_t t = 0; if (lane == 0) t = atomicAdd(tickets + (xcd * CTRL_SUBRANGES + sub) * CTRL_STRIDE_U32, 1u); return t; };
	// a returned ticket -> position in the cell list (>= hi: nothing is left anywhere on this XCD)
	auto resolve = [&](uint32_t pending) -> uint32_t {
		uint32_t t = readfirstlane_u32(pending);
		while (cur_lo + t >= cur_hi) {
			if (++used_up >= CTRL_SUBRANGES) return hi;
			sub = (sub + 1u) % CTRL_SUBRANGES; cur_lo = sub_begin(sub); cur_hi = sub_begin(sub + 1u);
			t = readfirstlane_u32(take());
		}
		return cur_lo + t;
	};
	auto entry = [&](uint32_t first) { return cell_list[first < hi ? first : lo]; };   // uniform address; clamped, never out of range
	uint2 rej = make_uint2(0u, 0u);
	uint32_t rej_n = 0;
	auto flush_rejects = [&]() {
		uint32_t hb = 0;
		if (lane == 0) hb = atomicAdd(reject_count, rej_n);
		hb = readfirstlane_u32(hb);
		if ((uint32_t)lane < rej_n) reject_list[hb + (uint32_t)lane] = rej;
	};
	uint32_t tk_pending = take();
	uint32_t first_cur = resolve(tk_pending);
	if (first_cur >= hi) return;
	tk_pending = take();
	uint2 oc = entry(first_cur);
	uint32_t first_next = resolve(tk_pending);
	tk_pending = take();
	uint2 oc_next = entry(first_next);
	uint32_t key = readfirstlane_u32(oc.y), p0 = readfirstlane_u32(oc.x);
	uint32_t s, e;
	const uint32_t nb_pack = neighbour_pack(lane);
	lookup_cell_packed(a, key, true, nb_pack, s, e);
	uint2 qrange = a.table_i[key];

	for (;;) {
		// ---- advance the pipeline: ticket of cell +2 has arrived, entry of cell +1 has arrived
		const uint32_t first_next2 = resolve(tk_pending);
		tk_pending = take();
		const bool have_next = first_next < hi;
		const uint32_t key_next = readfirstlane_u32(oc_next.y), p0_next = readfirstlane_u32(oc_next.x);
		oc_next = entry(first_next2);

		// ---- current cell: merge the x-triples of its 27 lookups into 9 runs
		// (four ds_bpermute, 24 cycles each by tools/ubench/valu_rate.hip; wave_shl:1 DPP moves in their place measured 3 % SLOWER
		//  on the whole query in three interleaved A/Bs of rotating order, profiles/r3_query_ab_micro.txt)
		const uint32_t s1 = __shfl_down(s, 1, WAVE), e1 = __shfl_down(e, 1, WAVE);
		const uint32_t s2 = __shfl_down(s, 2, WAVE), e2 = __shfl_down(e, 2, WAVE);
		RunRef RR;
		RR.run_start = (e > s) ? s : ((e1 > s1) ? s1 : s2);
		const uint32_t run_end = (e2 > s2) ? e2 : ((e1 > s1) ? e1 : e);
		RR.run_len = ((e > s) || (e1 > s1) || (e2 > s2)) ? run_end - RR.run_start : 0u;   // empty entries may hold any (s,s)
#if TNSX_TOTAL_BY_DPP
		// (the number of candidates = the lengths of the nine runs, which sit in lanes 0, 3, ..., 24: a masked sum over two rows of lanes instead of
		//  eighteen v_readlane and as many scalar instructions -- the cell bodies extract the runs themselves)
		RR.total = wave_sum32_masked(RR.run_len, 0x1249249ull);
#else
		{
			const Runs R0 = extract_runs(RR.run_start, RR.run_len);
			RR.total = R0.total;
		}
#endif
		const uint2 cur_q = qrange;
		// ---- lookups of the next cell: in flight while this one is processed
		{
#if TNSX_LANE_OPAQUE
			// (unpacked here, once per cell: see lookup_cell_packed.  The one instantiation that is still a register short at five waves -- per-point radii,
			//  symmetric, two different sets -- does not even keep the packed word: it derives it from the lane number again, eight instructions per cell)
			uint32_t pk;
			if (VARIABLE && SYM && !SELF && !FAT) { int l2 = lane; asm volatile("" : "+v"(l2)); pk = neighbour_pack(l2); }
			else { pk = nb_pack; asm volatile("" : "+v"(pk)); }
#else
			const uint32_t pk = nb_pack;
#endif
			lookup_cell_packed(a, key_next, have_next, pk, s, e);
		}
		qrange = a.table_i[have_next ? key_next : key];

		const uint32_t nq = cur_q.y - cur_q.x;
		bool pass_on = RR.total > 2u * (uint32_t)Q_SLOTS || nq > (uint32_t)WAVE;
		bool fat_culled = false;
#if TNSX_LANE_OPAQUE
		// The cell bodies use slot numbers k * 64 + lane for a dozen values of k (the validity of a last chunk, the survivors of the cull).  Left alone the
		// compiler computes all of them once, in front of this loop, and keeps them in a dozen VGPRs for the life of the kernel -- at the 96 registers of five
		// waves per SIMD the rest then spills (48 bytes per lane with per-point radii, 16 .. 64 for pairs of two sets).  A lane number that is opaque per cell makes
		// them what they are: one v_or_b32 where they are used.  (round 5; scratch 0 for every first-tier instantiation)
		int lane_c = lane;
		asm volatile("" : "+v"(lane_c));
#else
		const int lane_c = lane;
#endif
		if (!pass_on && !FAT && RR.total > (uint32_t)(VARIABLE ? TNSX_CULL_FROM_VARIABLE : TNSX_CULL_FROM)) {
			// more candidates than the loop holds: cull them against the bounding box of the query points; the cell is done
			// here if at most 512 survive
			pass_on = !(TNSX_CULL && fast_cell_culled<ARITH, VARIABLE, SYM, SELF, false>(a, RR, lane_c, cur_q, ps, wave_hits, my_slots));
		}
		else if (!pass_on && FAT && TNSX_CULL && TNSX_FAT_CULL && RR.total > (uint32_t)Q_SLOTS) {
			fat_culled = fast_cell_culled<ARITH, VARIABLE, SYM, SELF, true>(a, RR, lane_c, cur_q, ps, wave_hits, my_slots);
		}
		if (pass_on) {
			// not for this tier: goes to the next tier's worklist.  Collected one entry per lane and appended 64 at a time: the
			// worklist length is ONE counter, and e.g. a dense column of fluid sends most of its cells here.
			if ((uint32_t)lane == rej_n) rej = make_uint2(p0, key);
			if (++rej_n == (uint32_t)WAVE) { flush_rejects(); rej_n = 0; }
		}
		else if (!FAT && RR.total > (uint32_t)(VARIABLE ? TNSX_CULL_FROM_VARIABLE : TNSX_CULL_FROM)) { /* done by the culled path above */ }
		else if (fat_culled) { /* done by the culled path above */ }
		else if (RR.total == 0u && a.shared_empty != 0u) {
			// no candidate at all, and the offsets of this pair were pre-set to the shared empty record: nothing to do.  (The fluid of an
			// SPH scene searched in its boundary: most fluid cells are nowhere near it.)
		}
		else if (RR.total == 0u) {
			// no candidate at all (set_j is another, sparser or empty set): nq empty records, one int each
			const uint32_t qs0 = cur_q.x + ((uint32_t)lane_c < nq ? (uint32_t)lane_c : 0u);
			const uint32_t qi0 = a.orig_i ? a.orig_i[qs0] : __float_as_uint(a.xyzi_i[qs0].w);
			const uint32_t nqv = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((uint32_t)lane_c < nq && qi0 < a.query_limit));   // (a prefix, see fast_query_loop)
			bool okz;
			const uint64_t off = pool_alloc<FAT>(a, ps, nqv, lane_c, okz);
			if ((uint32_t)lane_c < nqv && okz) {
				a.records[off + lane_c] = 0;
				a.offs_by_orig[qi0] = off + lane_c;
			}
		}
		else fast_cell_nc<ARITH, VARIABLE, SYM, SELF, FAT>(a, RR, lane_c, cur_q, ps, wave_hits);

		if (!have_next) break;
		key = key_next; p0 = p0_next;
		first_next = first_next2;
	}
	if (rej_n) flush_rejects();
	pool_wave_done<FAT>(a, ps, wave_hits, lane);
}

// =====================================================================================================
// Candidate-presence filter of a pool pass over two DIFFERENT sets.  A fluid searched in its boundary: most fluid cells are nowhere
// near a boundary point, their lists are the shared empty record (see launch_shared_empty_begin), and walking them through the
// query pipeline -- a wave, a ticket, 27 lookups per cell -- is all the pass would do there (C3: 0.85 ms for the 0->1 pair).  The
// query cells that have any candidate are compacted into the worklist the query kernels then walk instead of the whole
// occupied-cell list.
// =====================================================================================================
// Two steps, both cheap because they start from the SPARSER side: (1) every occupied cell of the candidate set marks its 27
// neighbour cells in a byte map of the grid (k_mark_cells; the map is all zero between runs: the same kernel un-marks after the
// pass); (2) every occupied query cell looks at ONE byte and the marked ones are compacted (k_filter_marked).  (A first version let
// every query cell sum its 27 table entries itself: 90 us at C3 for 15 M scattered 8-byte loads.)
__global__ void __launch_bounds__(256) k_mark_cells(const uint2* __restrict__ occ_j, const uint32_t* __restrict__ n_occ_p, GridParams g, unsigned char* __restrict__ map,
                                                   unsigned char value)
{
	const uint32_t n_occ = *n_occ_p;
	const int nx = g.nx, ny = g.ny, nz = g.nz;
	// 32 threads per cell: 27 of them write one neighbour each (x fastest: three consecutive bytes per row)
	const uint32_t per_block = 256u / 32u;
	const uint32_t sub = threadIdx.x & 31u;
	for (uint32_t c = blockIdx.x * per_block + (threadIdx.x >> 5); c < n_occ; c += gridDim.x * per_block) {
		if (sub >= 27u) continue;
		const uint32_t key = occ_j[c].y;
		const int x = (int)(key % (uint32_t)nx) + (int)(sub % 3u) - 1;
		const int y = (int)((key / (uint32_t)nx) % (uint32_t)ny) + (int)((sub / 3u) % 3u) - 1;
		const int z = (int)(key / ((uint32_t)nx * (uint32_t)ny)) + (int)(sub / 9u) - 1;
		if (x >= 0 && x < nx && y >= 0 && y < ny && z >= 0 && z < nz) map[((size_t)z * ny + y) * nx + x] = value;
	}
}
// (one atomic on *n_out per 2048 cells: one per wave and 64 cells -- 8.8 k atomics on one cache line at C3 -- made this an 80 us kernel)
__global__ void __launch_bounds__(256) k_filter_marked(const uint2* __restrict__ occ, const uint32_t* __restrict__ n_occ_p, const unsigned char* __restrict__ map,
                                                      uint2* __restrict__ out, uint32_t* __restrict__ n_out)
{
	constexpr int PER = 8;
	__shared__ uint32_t s_wave[4], s_base;
	const uint32_t n_occ = *n_occ_p;
	const uint32_t w = threadIdx.x / WAVE;
	for (uint32_t chunk = blockIdx.x * 256u * PER; chunk < n_occ; chunk += gridDim.x * 256u * PER) {
		uint2 oc[PER];
		uint64_t m[PER];
		uint32_t mine = 0;
		#pragma unroll
		for (int k = 0; k < PER; k++) {
			const uint32_t i = chunk + (uint32_t)k * 256u + threadIdx.x;
			oc[k] = i < n_occ ? occ[i] : make_uint2(0u, 0u);
		}
		#pragma unroll
		for (int k = 0; k < PER; k++) {
			const uint32_t i = chunk + (uint32_t)k * 256u + threadIdx.x;
			const bool any = i < n_occ && map[oc[k].y] != 0;
			m[k] = __builtin_amdgcn_ballot_w64(any);
			mine += (uint32_t)__popcll(m[k]);
		}
		if (lane_id() == 0) s_wave[w] = mine;
		__syncthreads();
		if (threadIdx.x == 0) {
			const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
			s_base = total ? atomicAdd(n_out, total) : 0u;
		}
		__syncthreads();
		uint32_t off = s_base;
		for (uint32_t q = 0; q < w; q++) off += s_wave[q];
		#pragma unroll
		for (int k = 0; k < PER; k++) {
			if ((m[k] >> lane_id()) & 1ull) out[off + mbcnt64(m[k])] = oc[k];
			off += (uint32_t)__popcll(m[k]);
		}
		__syncthreads();
	}
}
void launch_mark_cells(const uint2* occ_j, const uint32_t* n_occ_j, GridParams g, unsigned char* map, unsigned char value, size_t max_cells_j, hipStream_t s)
{
	size_t blocks = (max_cells_j + 7) / 8;
	blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
	hipLaunchKernelGGL(k_mark_cells, dim3((unsigned)blocks), dim3(256), 0, s, occ_j, n_occ_j, g, map, value);
}
void launch_filter_marked(const uint2* occ_i, const uint32_t* n_occ_i, const unsigned char* map, uint2* out, uint32_t* n_out, size_t max_cells, hipStream_t s)
{
	size_t blocks = (max_cells + 2047) / 2048;
	blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
	hipLaunchKernelGGL(k_filter_marked, dim3((unsigned)blocks), dim3(256), 0, s, occ_i, n_occ_i, map, out, n_out);
}

template <int ARITH, bool VARIABLE, bool SYM, bool SELF, int MODE>
static void launch_query_t(const QueryArgs& a, int blocks, hipStream_t s)
{
	if (a.blk_j) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query<ARITH, VARIABLE, SYM, SELF, MODE, true>), dim3(blocks), dim3(Q_THREADS), 0, s, a);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query<ARITH, VARIABLE, SYM, SELF, MODE, false>), dim3(blocks), dim3(Q_THREADS), 0, s, a);
}
template <int ARITH, bool VARIABLE, bool SYM, bool SELF>
static void launch_pool_t(const QueryArgs& a, int blocks_fast, int blocks_heavy, int tiers, hipStream_t s)
{
	// three tiers: fast kernel (<= 512 candidates per cell) over all occupied cells -> fat kernel (<= 1024) over its rejects ->
	// general kernel over what is left (more candidates, > 64 query points per cell, ...)
	if (tiers & 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_pool_fast<ARITH, VARIABLE, SYM, SELF, false>), dim3(blocks_fast), dim3(Q_THREADS), 0, s, a);
	if (!(tiers & 2)) return;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_pool_fast<ARITH, VARIABLE, SYM, SELF, true>), dim3(blocks_fast), dim3(Q_THREADS), 0, s, a);
	QueryArgs h = a;
	h.occ_i = a.heavy2;
	h.n_occ_i = a.n_heavy2;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query<ARITH, VARIABLE, SYM, SELF, MODE_POOL>), dim3(blocks_heavy), dim3(Q_THREADS), 0, s, h);
}
template <int ARITH, bool VARIABLE, bool SYM, bool SELF>
static void launch_query_3(const QueryArgs& a, const QueryConfig& c, int n_cus, hipStream_t s)
{
	const int per_cu = (c.blocks_per_cu >= 1 && c.blocks_per_cu <= 16) ? c.blocks_per_cu : 7;             // (tnsx_options.query_blocks_per_cu)
	// a multiple of 8 workgroups so that every XCD gets the same number (workgroup b runs on XCD b % 8)
	const int blocks = ((n_cus * per_cu + 7) / 8) * 8;
	if (c.mode == QUERY_COUNT) launch_query_t<ARITH, VARIABLE, SYM, SELF, MODE_COUNT>(a, blocks, s);
	else if (c.mode == QUERY_FILL) launch_query_t<ARITH, VARIABLE, SYM, SELF, MODE_FILL>(a, blocks, s);
	else if (a.blk_j) {
		// sparse grid: the general kernel over ALL occupied cells (it takes any cell; its records come from the common region of the pool)
		if (c.tiers & 1) launch_query_t<ARITH, VARIABLE, SYM, SELF, MODE_POOL>(a, blocks, s);
	}
	else {
		const int fast_per_cu = (c.fast_blocks_per_cu >= 1 && c.fast_blocks_per_cu <= 16) ? c.fast_blocks_per_cu : 8;   // (tnsx_options.fast_blocks_per_cu)
		launch_pool_t<ARITH, VARIABLE, SYM, SELF>(a, ((n_cus * fast_per_cu + 7) / 8) * 8, ((n_cus * 2 + 7) / 8) * 8, c.tiers, s);
	}
}
template <int ARITH, bool VARIABLE, bool SYM>
static void launch_query_2(const QueryArgs& a, const QueryConfig& c, int n_cus, hipStream_t s)
{
	if (c.self) launch_query_3<ARITH, VARIABLE, SYM, true>(a, c, n_cus, s); else launch_query_3<ARITH, VARIABLE, SYM, false>(a, c, n_cus, s);
}
template <int ARITH>
static void launch_query_1(const QueryArgs& a, const QueryConfig& c, int n_cus, hipStream_t s)
{
	if (!c.variable) launch_query_2<ARITH, false, false>(a, c, n_cus, s);
	else if (c.symmetric) launch_query_2<ARITH, true, true>(a, c, n_cus, s);
	else launch_query_2<ARITH, true, false>(a, c, n_cus, s);
}
void launch_query(const QueryArgs& a, const QueryConfig& c, int n_compute_units, hipStream_t s)
{
#ifdef TNSX_WITH_GROUP_FORMULATION   // (tools/ubench/tnsx_query_group.hip: a refuted formulation, linked only into the variant library of tools/build_group_variant.sh)
	if (c.mode == QUERY_POOL && c.groups && !c.variable) { launch_query_groups(a, c, n_compute_units, s); return; }
#endif
	if (c.arith == 0) launch_query_1<0>(a, c, n_compute_units, s); else launch_query_1<1>(a, c, n_compute_units, s);
}

}  // namespace tnsx
