// Spatial-slab layer behind the C ABI (include/tnsx.h, "slab layer"): one process (or thread) per GPU, slabs along x, ONE ghost-halo
// exchange per step over RCCL (ncclSend / ncclRecv inside one group -- xGMI is point to point, only the two neighbours of a slab
// ever talk to it), balanced cuts from an all-reduced x histogram.  No counterpart in the single-process reference
// (SURVEY.md section 8e); the search itself is the unchanged engine of tnsx_engine.cpp, driven through its public entry points:
// the ghosts are APPENDED to the owned points of their set, marked candidates-only (tnsx_set_query_count) and carry their global
// ids (tnsx_set_point_ids), so the lists the engine writes are global ids and no collective touches the data path.
//
// Wire format per neighbour, set and step: rows of W = 5 (+1 with per-point radii) floats [x, y, z, (r,) gid_lo, gid_hi]; row 0
// is a header whose first word is the row count.
//   exact step        (first step, after an overflow): counts first (4 bytes each way), then exactly the rows -- two rounds.
//   speculative step  (capacities known): ONE round of fixed-capacity messages, nothing is read on the host; rows past the count
//                     become NaN points on the receiver (the engine ignores them).  The counts are checked after the search
//                     (which synchronises anyway); an overflowed LINK is repaired by its two ends alone -- both see the same
//                     two numbers, so they agree without any collective -- and only they search again.
// RCCL is loaded at run time (dlopen): a single-GPU user of libtnsx.so needs no RCCL at all.
#include "tnsx.h"
#include "tnsx_kernels.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <thread>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
// (tnsx_engine.cpp) the stream / device a context works on
void* tnsx_internal_stream(tnsx_context* c);
int tnsx_internal_device(tnsx_context* c);
void tnsx_internal_set_sync_timeout(tnsx_context* c, double seconds);
}

namespace {

// ---------------------------------------------------------------------------------------------------------------- RCCL, loaded lazily
struct RcclApi {
	void* lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional
	ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;   // optional
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
	std::string error;
	bool load()
	{
		if (lib) return true;
		// the copy the process already has (PyTorch ships its own) first, then the ROCm installation's
		const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
		for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
		if (!lib) for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
		if (!lib) { error = std::string("RCCL not found (dlopen librccl.so): ") + (dlerror() ? dlerror() : ""); return false; }
#define TNSX_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); if (!field) { error = std::string("RCCL symbol missing: ") + name; lib = nullptr; return false; }
		TNSX_SYM(GetUniqueId, "ncclGetUniqueId") TNSX_SYM(CommInitRank, "ncclCommInitRank") TNSX_SYM(CommDestroy, "ncclCommDestroy")
		TNSX_SYM(GroupStart, "ncclGroupStart") TNSX_SYM(GroupEnd, "ncclGroupEnd") TNSX_SYM(Send, "ncclSend") TNSX_SYM(Recv, "ncclRecv")
		TNSX_SYM(AllReduce, "ncclAllReduce") TNSX_SYM(GetErrorString, "ncclGetErrorString")
#undef TNSX_SYM
		CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
		CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));
		return true;
	}
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

struct RcclTransport {
	ncclComm_t comm = nullptr;
	int device = 0;
};

int rccl_exchange(void* user, int, int, const tnsx_slab_op* ops, int n_ops, void* stream)
{
	RcclTransport* t = static_cast<RcclTransport*>(user);
	hipStream_t s = static_cast<hipStream_t>(stream);
	bool any = false;
	for (int k = 0; k < n_ops; k++) any = any || ops[k].send_bytes || ops[k].recv_bytes;
	if (!any) return 0;
	if (g_rccl.GroupStart() != ncclSuccess) return 1;
	int rc = 0;
	for (int k = 0; k < n_ops && rc == 0; k++) {
		if (ops[k].send_bytes && g_rccl.Send(ops[k].send, ops[k].send_bytes, ncclChar, ops[k].peer, t->comm, s) != ncclSuccess) rc = 1;
		if (ops[k].recv_bytes && g_rccl.Recv(ops[k].recv, ops[k].recv_bytes, ncclChar, ops[k].peer, t->comm, s) != ncclSuccess) rc = 1;
	}
	if (g_rccl.GroupEnd() != ncclSuccess) rc = 1;
	return rc;
}
int rccl_allreduce(void* user, int, int, void* buf, int count, int op, void* stream)
{
	RcclTransport* t = static_cast<RcclTransport*>(user);
	const ncclDataType_t dt = op == TNSX_SLAB_SUM_U32 ? ncclUint32 : ncclFloat32;
	const ncclRedOp_t ro = op == TNSX_SLAB_SUM_U32 ? ncclSum : (op == TNSX_SLAB_MIN_F32 ? ncclMin : ncclMax);
	return g_rccl.AllReduce(buf, buf, (size_t)count, dt, ro, t->comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}
void rccl_release(void* user)
{
	RcclTransport* t = static_cast<RcclTransport*>(user);
	if (t && t->comm) (void)g_rccl.CommDestroy(t->comm);
	delete t;
}
void rccl_abort(void* user)
{
	// the watchdog's last resort: the pending send / recv kernels of this rank return, the communicator is gone
	RcclTransport* t = static_cast<RcclTransport*>(user);
	if (t && t->comm && g_rccl.CommAbort) { (void)g_rccl.CommAbort(t->comm); t->comm = nullptr; }
}

// ---------------------------------------------------------------------------------------------------------------- in-process transport
// All slabs of a decomposition inside ONE process (a thread per slab, any streams, one or several devices that can reach each
// other's memory): what tests/test_gpu_slabs.py runs on a single GPU.  A message is handed over as {pointer, bytes, event}: the
// receiver orders its stream behind the sender's event and copies device to device; the sender orders its stream behind the copy.
// The state of one posted message is owned jointly by the queue entry / the receiver that took it and by the sender (round-5 advice: a sender that gave up
// after a timeout used to leave a receiver in mid-copy with pointers into its dead stack frame).  The events die with the last owner.
struct LocalSent {
	hipEvent_t ready = nullptr, done = nullptr;
	bool consumed = false;
	~LocalSent() { if (ready) (void)hipEventDestroy(ready); if (done) (void)hipEventDestroy(done); }
};
struct LocalMsg { const void* ptr; size_t bytes; std::shared_ptr<LocalSent> st; };
struct LocalGroup {
	int world = 0;
	std::mutex mu;
	std::condition_variable cv;
	std::vector<std::deque<LocalMsg>> box;     // [src * world + dst]
	// all-reduce
	int ar_arrived = 0, ar_left = 0;
	uint64_t ar_gen = 0;
	std::vector<uint32_t> ar_acc;
	bool ar_failed = false;                    // a rank withdrew from a reduction after a timeout: its contribution is in ar_acc, the group's reductions are void from here on
	int refs = 0;
	double host_wait_s = 120.0;                // bound of every host-side wait for a peer (tnsx_slab_set_watchdog of any slab on this group sets it; read and written under mu)
};
struct LocalTransport { LocalGroup* g; int rank; };

int local_exchange(void* user, int rank, int world, const tnsx_slab_op* ops, int n_ops, void* stream)
{
	LocalTransport* t = static_cast<LocalTransport*>(user);
	LocalGroup* g = t->g;
	hipStream_t s = static_cast<hipStream_t>(stream);
	std::vector<std::shared_ptr<LocalSent>> sent((size_t)n_ops);
	// 1. post every send
	for (int k = 0; k < n_ops; k++) {
		if (!ops[k].send_bytes) continue;
		auto e = std::make_shared<LocalSent>();
		if (hipEventCreateWithFlags(&e->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(e->ready, s) != hipSuccess) return 1;
		sent[(size_t)k] = e;
		std::lock_guard<std::mutex> lk(g->mu);
		g->box[(size_t)rank * world + ops[k].peer].push_back({ ops[k].send, ops[k].send_bytes, e });
		g->cv.notify_all();
	}
	// (a wait that timed out: the messages this rank posted and nobody took are withdrawn; one a peer has taken stays alive through the peer's reference)
	auto give_up = [&]() {
		std::lock_guard<std::mutex> lk(g->mu);
		for (int k = 0; k < n_ops; k++) {
			if (!sent[(size_t)k]) continue;
			auto& q = g->box[(size_t)rank * world + ops[k].peer];
			for (auto it = q.begin(); it != q.end();) { if (it->st == sent[(size_t)k]) it = q.erase(it); else ++it; }
		}
		return 3;   // timed out
	};
	// 2. take every message addressed to this rank, in the order of the ops
	int rc = 0;
	for (int k = 0; k < n_ops; k++) {
		if (!ops[k].recv_bytes) continue;
		LocalMsg m;
		{
			std::unique_lock<std::mutex> lk(g->mu);
			auto& q = g->box[(size_t)ops[k].peer * world + rank];
			// (bounded: a peer that never posts its message must not hang this thread for ever -- the slab layer's claim that every wait of a step is bounded)
			if (!g->cv.wait_for(lk, std::chrono::duration<double>(g->host_wait_s), [&] { return !q.empty(); })) { lk.unlock(); return give_up(); }
			m = q.front();
			q.pop_front();
		}
		hipEvent_t done = nullptr;
		if (m.bytes != ops[k].recv_bytes) rc = 2;   // the two ends disagree on a message size: a protocol error
		if (rc == 0 && (hipStreamWaitEvent(s, m.st->ready, 0) != hipSuccess ||
		                hipMemcpyAsync(ops[k].recv, m.ptr, m.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)) rc = 1;
		if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess || hipEventRecord(done, s) != hipSuccess) rc = rc ? rc : 1;
		std::lock_guard<std::mutex> lk(g->mu);
		m.st->done = done;
		m.st->consumed = true;
		g->cv.notify_all();
	}
	// 3. the send buffers may be reused once the receivers' copies are ordered before this stream's later work
	for (int k = 0; k < n_ops; k++) {
		if (!sent[(size_t)k]) continue;
		LocalSent& e = *sent[(size_t)k];
		{
			std::unique_lock<std::mutex> lk(g->mu);
			if (!g->cv.wait_for(lk, std::chrono::duration<double>(g->host_wait_s), [&] { return e.consumed; })) { lk.unlock(); return give_up(); }
		}
		if (e.done) { if (hipStreamWaitEvent(s, e.done, 0) != hipSuccess) rc = rc ? rc : 1; }
		// (both events are destroyed with the message's state, i.e. after this wait was enqueued and after the receiver let go of it)
	}
	return rc;
}
int local_allreduce(void* user, int, int world, void* buf, int count, int op, void* stream)
{
	LocalTransport* t = static_cast<LocalTransport*>(user);
	LocalGroup* g = t->g;
	hipStream_t s = static_cast<hipStream_t>(stream);
	std::vector<uint32_t> mine((size_t)count);
	if (hipMemcpyAsync(mine.data(), buf, (size_t)count * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 1;
	std::vector<uint32_t> result;
	{
		std::unique_lock<std::mutex> lk(g->mu);
		const auto bound = std::chrono::duration<double>(g->host_wait_s);
		if (g->ar_failed) return 3;                                                     // (a rank withdrew earlier: what ar_acc holds is not a reduction any more)
		if (!g->cv.wait_for(lk, bound, [&] { return g->ar_left == 0 || g->ar_failed; }) || g->ar_failed) return 3;   // the previous reduction has been read by everybody (3: timed out)
		if (g->ar_arrived == 0) g->ar_acc = mine;
		else for (int i = 0; i < count; i++) {
			uint32_t& a = g->ar_acc[(size_t)i];
			if (op == TNSX_SLAB_SUM_U32) a += mine[(size_t)i];
			else {
				float fa, fb; std::memcpy(&fa, &a, 4); std::memcpy(&fb, &mine[(size_t)i], 4);
				fa = op == TNSX_SLAB_MIN_F32 ? std::min(fa, fb) : std::max(fa, fb);
				std::memcpy(&a, &fa, 4);
			}
		}
		const uint64_t gen = g->ar_gen;
		if (++g->ar_arrived == world) { g->ar_arrived = 0; g->ar_left = world; g->ar_gen++; g->cv.notify_all(); }
		else if (!g->cv.wait_for(lk, bound, [&] { return g->ar_gen != gen || g->ar_failed; }) || g->ar_gen == gen) {
			// a rank that never came: this one withdraws.  Its contribution stays in ar_acc, so the generation can never complete correctly: the group's
			// reductions fail from here on (every waiter is woken and returns 3) instead of handing out a polluted sum to a late arrival.
			g->ar_failed = true;
			g->cv.notify_all();
			return 3;
		}
		result = g->ar_acc;
		if (--g->ar_left == 0) g->cv.notify_all();
	}
	if (hipMemcpyAsync(buf, result.data(), (size_t)count * 4, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 1;
	return 0;
}
void local_release(void* user)
{
	LocalTransport* t = static_cast<LocalTransport*>(user);
	if (!t) return;
	bool last;
	{ std::lock_guard<std::mutex> lk(t->g->mu); last = --t->g->refs == 0; }
	if (last) delete t->g;
	delete t;
}

// ---------------------------------------------------------------------------------------------------------------- the slab
struct DBuf {
	void* p = nullptr; size_t cap = 0;
	~DBuf() { if (p) (void)hipFree(p); }
	bool reserve(size_t bytes, bool keep = false)
	{
		if (bytes <= cap) return true;
		void* q = nullptr;
		const size_t want = bytes + bytes / 8 + 4096;
		if (hipMalloc(&q, want) != hipSuccess) return false;
		if (keep && p && cap) (void)hipMemcpy(q, p, cap, hipMemcpyDeviceToDevice);
		if (p) (void)hipFree(p);
		p = q; cap = want;
		return true;
	}
	template <typename T> T* as() const { return static_cast<T*>(p); }
	// after a watchdog timeout on a transport that cannot abort: the stream may still read or write this memory whenever it drains (and hipFree would block on
	// it): the buffer is deliberately leaked instead of freed under the stream's feet
	void abandon() { p = nullptr; cap = 0; }
};
struct SetState {
	int set_id = -1;
	DBuf send[2], recv[2];        // rows of W floats behind one header row
	uint32_t cap_s[2] = { 0, 0 }, cap_r[2] = { 0, 0 };
	bool caps_known[2] = { false, false };
	DBuf xyz, radii, ids;         // [owned | ghosts]
	int n_owned = 0, n_ghost = 0;
	uint32_t n_out[2] = { 0, 0 }, n_in[2] = { 0, 0 };
};
inline uint32_t capacity_rule(uint32_t count) { return count + count / 4 + 256; }

}  // namespace

struct tnsx_slab {
	tnsx_context* engine = nullptr;
	tnsx_slab_transport tr{};
	int rank = 0, world = 1;
	float lo = 0, hi = 0, radius = -1, max_radius = 0, halo = 0;
	bool variable = false, speculative = true;
	hipStream_t stream = nullptr;
	int device = 0;
	std::vector<SetState> sets;
	std::vector<std::pair<std::pair<int, int>, int>> active;   // ((i, j), on) in terms of slab set indices
	bool active_applied = false;
	DBuf d_small;                 // device scratch: pack counts (2 per set), flags
	unsigned int* h_small = nullptr;   // pinned mirror
	size_t small_words = 0;
	tnsx_slab_info info{};
	std::string last_error;
	double watchdog_s = 120.0;    // bound of every wait on the stream (tnsx_slab_set_watchdog)
	bool collect_times = false;   // tnsx_slab_set_collect_times: an event pair around every exchange round, read at the end of the step
	hipEvent_t ev[2][3] = { { nullptr, nullptr, nullptr }, { nullptr, nullptr, nullptr } };   // [begin | end][round of the step]
	int ev_rounds = 0;
};

namespace {
tnsx_status sfail(tnsx_slab* s, tnsx_status st, const char* fmt, ...)
{
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	s->last_error = buf;
	return st;
}
#define SHIP(s, call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return sfail(s, TNSX_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)
#define SENG(s, call) do { const tnsx_status r_ = (call); if (r_ != TNSX_OK) return sfail(s, r_, "%s: %s", #call, tnsx_last_error((s)->engine)); } while (0)

// hipStreamSynchronize with a deadline.  -> 0 ok, 1 HIP error, 2 timed out
int wait_stream(hipStream_t st, double seconds)
{
	if (!(seconds > 0.0)) return hipStreamSynchronize(st) == hipSuccess ? 0 : 1;
	const auto t0 = std::chrono::steady_clock::now();
	for (;;) {
		const hipError_t q = hipStreamQuery(st);
		if (q == hipSuccess) return 0;
		if (q != hipErrorNotReady) return 1;
		if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return 2;
		std::this_thread::yield();
	}
}
inline int peer_of(const tnsx_slab* s, int side) { return side == 0 ? s->rank - 1 : s->rank + 1; }
// the watchdog fired: say which links this rank was waiting on, make the transport let go, fail the step
tnsx_status timed_out(tnsx_slab* s, const char* what)
{
	std::string links;
	for (int side = 0; side < 2; side++) {
		const int p = peer_of(s, side);
		if (p < 0 || p >= s->world) continue;
		char b[160];
		const SetState* st = s->sets.empty() ? nullptr : &s->sets[0];
		std::snprintf(b, sizeof b, "%s link %d <-> %d (set 0: %u rows agreed out, %u in)", links.empty() ? "" : ";", s->rank, p, st ? st->cap_s[side] : 0u, st ? st->cap_r[side] : 0u);
		links += b;
	}
	if (s->tr.abort) s->tr.abort(s->tr.user);
	return sfail(s, TNSX_ERR_TIMEOUT, "rank %d of %d: the stream did not drain within %.1f s while %s -- a neighbour that never posted its side of the exchange, or a message-size "
	             "mismatch.%s.  The transport was %s", s->rank, s->world, s->watchdog_s, what, links.c_str(), s->tr.abort ? "aborted" : "left as it is (no abort hook)");
}
#define SWAIT(s, what) do { const int w_ = wait_stream((s)->stream, (s)->watchdog_s); if (w_ == 2) return timed_out(s, what); if (w_ == 1) return sfail(s, TNSX_ERR_HIP, "HIP error while %s", what); } while (0)
inline bool has_side(const tnsx_slab* s, int side) { const int p = peer_of(s, side); return p >= 0 && p < s->world; }

// scratch layout (32-bit words): [set * 8 + 0..1] pack counts, [+2..3] received header counts, [+4] radius flag, [+5] id flag
inline unsigned int* small_dev(tnsx_slab* s, size_t k) { return s->d_small.as<unsigned int>() + k * 8; }
inline unsigned int* small_host(tnsx_slab* s, size_t k) { return s->h_small + k * 8; }

tnsx_status ensure_small(tnsx_slab* s, size_t n_sets)
{
	if (n_sets * 8 <= s->small_words) return TNSX_OK;
	if (!s->d_small.reserve(n_sets * 8 * sizeof(unsigned int))) return sfail(s, TNSX_ERR_HIP, "out of device memory");
	if (s->h_small) (void)hipHostFree(s->h_small);
	SHIP(s, hipHostMalloc((void**)&s->h_small, n_sets * 8 * sizeof(unsigned int)));
	s->small_words = n_sets * 8;
	return TNSX_OK;
}

// pack the halo rows of one set; sides[] says which sides are wanted; cap_rows[] = rows the send buffers may take (without header)
tnsx_status pack(tnsx_slab* s, size_t k, const float* xyz, const long long* gids, const float* radii, int n, const bool want[2], const uint32_t cap_rows[2], bool wait_counts)
{
	SetState& st = s->sets[k];
	const size_t W = s->variable ? 6 : 5;
	for (int side = 0; side < 2; side++) if (want[side] && !st.send[side].reserve(((size_t)cap_rows[side] + 1) * W * 4, false)) return sfail(s, TNSX_ERR_HIP, "out of device memory (halo buffers)");
	unsigned int host_counts[2] = { 0, 0 };
	SENG(s, tnsx_halo_pack(s->engine, xyz, radii, gids, n, s->lo + s->halo, s->hi - s->halo,
	                       want[0] ? st.send[0].as<float>() + W : nullptr, want[1] ? st.send[1].as<float>() + W : nullptr, cap_rows[0], cap_rows[1],
	                       small_dev(s, k), wait_counts ? host_counts : nullptr));
	if (wait_counts) for (int side = 0; side < 2; side++) if (want[side]) st.n_out[side] = host_counts[side];
	// header word = the count, device to device
	for (int side = 0; side < 2; side++) if (want[side]) SHIP(s, hipMemcpyAsync(st.send[side].p, small_dev(s, k) + side, 4, hipMemcpyDeviceToDevice, s->stream));
	return TNSX_OK;
}

// [owned | ghosts] of one set -> the engine.  counts_on_device: the ghost rows of side `side` that exist are given by the header word
// of its receive buffer (rows past it become NaN points); else by n_in (exact).
tnsx_status assemble(tnsx_slab* s, size_t k, const float* xyz, const long long* gids, const float* radii, int n, const uint32_t rows[2], bool counts_on_device)
{
	SetState& st = s->sets[k];
	const size_t W = s->variable ? 6 : 5;
	const size_t m = (size_t)rows[0] + rows[1];
	const size_t total = (size_t)n + m;
	if (total > 0x7fffffffull) return sfail(s, TNSX_ERR_LIST_TOO_LONG, "a slab holds more than 2^31 - 1 points");
	if (!st.xyz.reserve(std::max<size_t>(total, 1) * 12) || !st.ids.reserve(std::max<size_t>(total, 1) * 4) || (s->variable && !st.radii.reserve(std::max<size_t>(total, 1) * 4)))
		return sfail(s, TNSX_ERR_HIP, "out of device memory (slab point buffers)");
	if (n > 0) {
		if (xyz != st.xyz.as<float>()) SHIP(s, hipMemcpyAsync(st.xyz.p, xyz, (size_t)n * 12, hipMemcpyDeviceToDevice, s->stream));
		if (s->variable) SHIP(s, hipMemcpyAsync(st.radii.p, radii, (size_t)n * 4, hipMemcpyDeviceToDevice, s->stream));
		tnsx::launch_slab_ids(gids, n, st.ids.as<int>(), small_dev(s, k) + 5, s->stream);
		if (s->variable) tnsx::launch_slab_flag_gt(radii, n, s->max_radius, small_dev(s, k) + 4, s->stream);
	}
	size_t at = (size_t)n;
	for (int side = 0; side < 2; side++) {
		if (!rows[side]) continue;
		const float* rbuf = st.recv[side].as<float>();
		tnsx::launch_slab_unpack(rbuf + W, rows[side], counts_on_device ? reinterpret_cast<const unsigned int*>(rbuf) : nullptr, (int)W,
		                         st.xyz.as<float>() + 3 * at, s->variable ? st.radii.as<float>() + at : nullptr, st.ids.as<int>() + at, small_dev(s, k) + 5, s->stream);
		at += rows[side];
	}
	SHIP(s, hipGetLastError());
	st.n_owned = n; st.n_ghost = (int)m;
	const unsigned flags = TNSX_F32 | TNSX_DEVICE | (s->variable ? TNSX_VARIABLE : 0u);
	const void* rp = s->variable ? st.radii.p : nullptr;
	if (st.set_id < 0) {
		const int id = tnsx_add_point_set(s->engine, st.xyz.p, rp, (int)total, flags);
		if (id < 0) return sfail(s, (tnsx_status)(-id), "tnsx_add_point_set: %s", tnsx_last_error(s->engine));
		st.set_id = id;
	}
	else SENG(s, tnsx_resize_point_set(s->engine, st.set_id, st.xyz.p, rp, (int)total, flags));
	SENG(s, tnsx_set_query_count(s->engine, st.set_id, n));
	SENG(s, tnsx_set_point_ids(s->engine, st.set_id, st.ids.as<int>()));
	return TNSX_OK;
}

tnsx_status run_engine(tnsx_slab* s)
{
	if (!s->active_applied) {
		if (s->active.empty()) s->active.push_back({ { 0, 0 }, 1 });
		for (const auto& a : s->active) {
			if ((size_t)a.first.first >= s->sets.size() || (size_t)a.first.second >= s->sets.size()) return sfail(s, TNSX_ERR_INVALID, "tnsx_slab_set_active_search: set %d or %d was never stepped", a.first.first, a.first.second);
			SENG(s, tnsx_set_active_search(s->engine, s->sets[(size_t)a.first.first].set_id, s->sets[(size_t)a.first.second].set_id, a.second));
		}
		s->active_applied = true;
	}
	tnsx_internal_set_sync_timeout(s->engine, s->watchdog_s);
	const tnsx_status r = tnsx_run(s->engine);
	tnsx_internal_set_sync_timeout(s->engine, 0.0);
	if (r == TNSX_ERR_TIMEOUT) return timed_out(s, "waiting for the halo exchange and the search behind it");
	if (r != TNSX_OK) return sfail(s, r, "tnsx_run: %s", tnsx_last_error(s->engine));
	return TNSX_OK;
}

int do_exchange(tnsx_slab* s, const std::vector<tnsx_slab_op>& ops)
{
	if (ops.empty() || !s->tr.exchange) return 0;
	s->info.rounds_last++;
	for (const tnsx_slab_op& o : ops) s->info.bytes_sent += o.send_bytes;
	// (stage pass of a benchmark: how long the messages of this round occupy the stream -- a step has at most three rounds)
	const int slot = s->collect_times && s->ev_rounds < 3 ? s->ev_rounds : -1;
	if (slot >= 0) {
		for (int b = 0; b < 2; b++) if (!s->ev[b][slot] && hipEventCreate(&s->ev[b][slot]) != hipSuccess) return 1;
		if (hipEventRecord(s->ev[0][slot], s->stream) != hipSuccess) return 1;
	}
	const int rc = s->tr.exchange(s->tr.user, s->rank, s->world, ops.data(), (int)ops.size(), s->stream);
	if (slot >= 0 && rc == 0) { if (hipEventRecord(s->ev[1][slot], s->stream) != hipSuccess) return 1; s->ev_rounds++; }
	return rc;
}
// what kind of transport a table of functions is, and how many ranks it spans as far as the library can tell
void describe_transport(const tnsx_slab_transport& tr, int world, tnsx_slab_info& info)
{
	info.transport_kind = !tr.exchange ? 0 : (tr.exchange == rccl_exchange ? 1 : (tr.exchange == local_exchange ? 2 : 3));
	info.transport_ranks = info.transport_kind == 0 ? 1 : -1;
	if (info.transport_kind == 1) {
		RcclTransport* t = static_cast<RcclTransport*>(tr.user);
		int n = -1;
		if (t && t->comm && g_rccl.CommCount && g_rccl.CommCount(t->comm, &n) == ncclSuccess) info.transport_ranks = n;
	}
	else if (info.transport_kind == 2) info.transport_ranks = world;
}

}  // namespace

extern "C" {

static thread_local std::string g_slab_create_error;   // what tnsx_slab_create / tnsx_slab_balanced_cuts had to say (no slab to carry it)
const char* tnsx_slab_last_error(const tnsx_slab* s) { return s ? s->last_error.c_str() : g_slab_create_error.c_str(); }

// ------------------------------------------------------------------------------------------------ transports
tnsx_status tnsx_slab_rccl_unique_id(void* out128)
{
	std::lock_guard<std::mutex> lk(g_rccl_mu);
	if (!out128 || !g_rccl.load()) return TNSX_ERR_STATE;
	ncclUniqueId id;
	if (g_rccl.GetUniqueId(&id) != ncclSuccess) return TNSX_ERR_HIP;
	std::memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
	return TNSX_OK;
}
const char* tnsx_slab_rccl_error(void) { return g_rccl.error.c_str(); }

tnsx_status tnsx_slab_transport_rccl(const void* unique_id128, int rank, int world, int device, tnsx_slab_transport* out)
{
	if (!unique_id128 || !out || rank < 0 || rank >= world) return TNSX_ERR_INVALID;
	{
		std::lock_guard<std::mutex> lk(g_rccl_mu);
		if (!g_rccl.load()) return TNSX_ERR_STATE;
	}
	if (device >= 0 && hipSetDevice(device) != hipSuccess) return TNSX_ERR_HIP;
	ncclUniqueId id;
	std::memcpy(id.internal, unique_id128, NCCL_UNIQUE_ID_BYTES);
	RcclTransport* t = new RcclTransport();
	t->device = device;
	const ncclResult_t r = g_rccl.CommInitRank(&t->comm, world, id, rank);
	if (r != ncclSuccess) { g_rccl.error = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); delete t; return TNSX_ERR_HIP; }
	out->user = t; out->exchange = rccl_exchange; out->allreduce = rccl_allreduce; out->release = rccl_release; out->abort = rccl_abort;
	return TNSX_OK;
}

tnsx_status tnsx_slab_local_group_create(int world, void** group_out)
{
	if (world < 1 || !group_out) return TNSX_ERR_INVALID;
	LocalGroup* g = new LocalGroup();
	g->world = world;
	g->box.resize((size_t)world * world);
	g->refs = 1;            // the creator's reference, dropped by tnsx_slab_local_group_release
	*group_out = g;
	return TNSX_OK;
}
void tnsx_slab_local_group_release(void* group)
{
	LocalGroup* g = static_cast<LocalGroup*>(group);
	if (!g) return;
	bool last;
	{ std::lock_guard<std::mutex> lk(g->mu); last = --g->refs == 0; }
	if (last) delete g;
}
tnsx_status tnsx_slab_transport_local(void* group, int rank, tnsx_slab_transport* out)
{
	LocalGroup* g = static_cast<LocalGroup*>(group);
	if (!g || !out || rank < 0 || rank >= g->world) return TNSX_ERR_INVALID;
	{ std::lock_guard<std::mutex> lk(g->mu); g->refs++; }
	LocalTransport* t = new LocalTransport{ g, rank };
	out->user = t; out->exchange = local_exchange; out->allreduce = local_allreduce; out->release = local_release; out->abort = nullptr;
	return TNSX_OK;
}
void tnsx_slab_transport_release(tnsx_slab_transport* t)
{
	if (t && t->release) t->release(t->user);
	if (t) { t->user = nullptr; t->exchange = nullptr; t->allreduce = nullptr; t->release = nullptr; t->abort = nullptr; }
}

// ------------------------------------------------------------------------------------------------ decomposition
tnsx_status tnsx_slab_balanced_cuts(tnsx_context* engine, const tnsx_slab_transport* tr, int rank, int world, int n_sets, const float* const* xyz,
                                    const int* n_points, float plane_width, int n_slabs, float* cuts_out)
{
	if (!engine || !cuts_out || n_sets < 0 || world < 1 || !(plane_width > 0.0f)) return TNSX_ERR_INVALID;
	// (cuts computed from the local points alone would differ from rank to rank: points dropped or owned twice)
	if (world > 1 && (!tr || !tr->allreduce)) { g_slab_create_error = "tnsx_slab_balanced_cuts: world > 1 needs a transport with an all-reduce"; return TNSX_ERR_INVALID; }
	if (n_slabs <= 0) n_slabs = world;
	const int device = tnsx_internal_device(engine);
	hipStream_t stream = static_cast<hipStream_t>(tnsx_internal_stream(engine));
	if (hipSetDevice(device) != hipSuccess) return TNSX_ERR_HIP;
	const int MAX_PLANES = 32768;   // the reference's cells-per-axis limit (TreeNSearch.cpp:510-515)
	DBuf d;
	if (!d.reserve((size_t)(MAX_PLANES + 8) * 4)) return TNSX_ERR_HIP;
	// ---- global x range: min / max over all sets and ranks
	float init[2] = { FLT_MAX, -FLT_MAX };
	if (hipMemcpyAsync(d.p, init, 8, hipMemcpyHostToDevice, stream) != hipSuccess) return TNSX_ERR_HIP;
	for (int k = 0; k < n_sets; k++) if (n_points[k] > 0) tnsx::launch_slab_x_range(xyz[k], n_points[k], d.as<float>(), stream);
	if (world > 1 && tr && tr->allreduce) {
		if (tr->allreduce(tr->user, rank, world, d.as<float>(), 1, TNSX_SLAB_MIN_F32, stream) || tr->allreduce(tr->user, rank, world, d.as<float>() + 1, 1, TNSX_SLAB_MAX_F32, stream)) return TNSX_ERR_HIP;
	}
	float range[2];
	if (hipMemcpyAsync(range, d.p, 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return TNSX_ERR_HIP;
	float x0 = range[0], x1 = range[1];
	if (!(std::isfinite(x0) && std::isfinite(x1)) || x1 < x0) { x0 = 0.0f; x1 = 0.0f; }   // no points anywhere
	const double planes_d = std::floor(((double)x1 - (double)x0) / (double)plane_width) + 1.0;   // (range-checked before the cast)
	if (!(planes_d <= (double)MAX_PLANES) || planes_d < (double)n_slabs) {
		g_slab_create_error = "tnsx_slab_balanced_cuts: the x range holds too many planes of plane_width (> 32768), or fewer planes than slabs";
		return TNSX_ERR_GRID_TOO_LARGE;
	}
	const int n_planes = (int)planes_d;
	// ---- histogram of the x planes, all ranks
	unsigned int* hist = d.as<unsigned int>() + 8;
	if (hipMemsetAsync(hist, 0, (size_t)n_planes * 4, stream) != hipSuccess) return TNSX_ERR_HIP;
	const float inv = 1.0f / plane_width;
	for (int k = 0; k < n_sets; k++) if (n_points[k] > 0 && tnsx_x_histogram(engine, xyz[k], n_points[k], x0, inv, n_planes, hist) != TNSX_OK) return TNSX_ERR_HIP;
	if (world > 1 && tr && tr->allreduce && tr->allreduce(tr->user, rank, world, hist, n_planes, TNSX_SLAB_SUM_U32, stream)) return TNSX_ERR_HIP;
	std::vector<unsigned int> h((size_t)n_planes);
	if (hipMemcpyAsync(h.data(), hist, (size_t)n_planes * 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return TNSX_ERR_HIP;
	// ---- cuts at the plane boundaries closest to the k / n_slabs quantiles, at least one plane per slab
	std::vector<uint64_t> cum((size_t)n_planes);
	uint64_t run = 0;
	for (int b = 0; b < n_planes; b++) { run += h[(size_t)b]; cum[(size_t)b] = run; }
	const uint64_t total = run;
	cuts_out[0] = -INFINITY; cuts_out[n_slabs] = INFINITY;
	int prev = 0;
	for (int k = 1; k < n_slabs; k++) {
		int b = k;
		if (total) {
			const double target = (double)total * k / n_slabs;
			b = (int)(std::lower_bound(cum.begin(), cum.end(), target, [](uint64_t c, double t) { return (double)c < t; }) - cum.begin()) + 1;   // boundary b has cum[b - 1] points to its left
			if (b >= 2 && std::fabs((double)cum[(size_t)b - 2] - target) <= std::fabs((double)cum[(size_t)std::min(b, n_planes) - 1] - target)) b -= 1;
		}
		b = std::min(std::max(b, prev + 1), n_planes - (n_slabs - k));
		cuts_out[k] = x0 + (float)b * plane_width;
		prev = b;
	}
	return TNSX_OK;
}

// ------------------------------------------------------------------------------------------------ the slab
tnsx_status tnsx_slab_create(tnsx_context* engine, const tnsx_slab_transport* transport, int rank, int world, float slab_lo, float slab_hi, float radius,
                             float max_radius, float halo_margin, int speculative, tnsx_slab** out)
{
	if (!engine || !out || world < 1 || rank < 0 || rank >= world) return TNSX_ERR_INVALID;
	if (world > 1 && (!transport || !transport->exchange)) return TNSX_ERR_INVALID;
	tnsx_slab* s = new tnsx_slab();
	s->engine = engine;
	if (transport) s->tr = *transport;
	s->rank = rank; s->world = world;
	s->lo = slab_lo; s->hi = slab_hi;
	s->variable = !(radius > 0.0f);
	s->radius = radius;
	s->max_radius = s->variable ? max_radius : radius;
	if (!(s->max_radius > 0.0f)) { delete s; return TNSX_ERR_INVALID; }   // per-point radii: an upper bound of every radius sizes the halo
	s->halo = s->max_radius * (1.0f + (halo_margin > 0.0f ? halo_margin : 1.0e-3f));
	s->speculative = speculative != 0;
	// ghosts only ever come from rank - 1 and rank + 1: a slab with two neighbours that is thinner than the halo would leave points of rank - 1
	// within the radius of points of rank + 1 unseen by either -- silently incomplete lists.  (tnsx_slab_balanced_cuts never cuts thinner than its
	// plane_width; caller-supplied cuts are checked here.)
	if (rank > 0 && rank < world - 1 && !(slab_hi - slab_lo >= s->halo)) {
		char b[256];
		std::snprintf(b, sizeof b, "tnsx_slab_create: slab %d of %d is %g wide, thinner than the halo %g (ghosts are exchanged with the two adjacent slabs only)", rank, world,
		              (double)(slab_hi - slab_lo), (double)s->halo);
		g_slab_create_error = b;
		delete s;
		return TNSX_ERR_INVALID;
	}
	s->stream = static_cast<hipStream_t>(tnsx_internal_stream(engine));
	s->device = tnsx_internal_device(engine);
	if (!s->stream) { delete s; return TNSX_ERR_STATE; }   // (a multi-device context shards host data itself: tnsx_options.n_devices)
	if (!s->variable && tnsx_set_search_radius(engine, radius) != TNSX_OK) { delete s; return TNSX_ERR_CONFIG; }
	describe_transport(s->tr, world, s->info);
	*out = s;
	return TNSX_OK;
}

void tnsx_slab_destroy(tnsx_slab* s)
{
	if (!s) return;
	(void)hipSetDevice(s->device);
	(void)hipStreamSynchronize(s->stream);
	// the engine holds the slab's [owned | ghosts] buffers as TNSX_DEVICE inputs of its sets: they become empty sets before the buffers go
	// (an engine that outlives its slab must not read freed memory at its next run / prepare_zsort)
	for (SetState& st : s->sets) {
		if (st.set_id < 0) continue;
		(void)tnsx_set_point_ids(s->engine, st.set_id, nullptr);
		(void)tnsx_resize_point_set(s->engine, st.set_id, nullptr, nullptr, 0, TNSX_F32 | TNSX_DEVICE | (s->variable ? TNSX_VARIABLE : 0u));
	}
	if (s->h_small) (void)hipHostFree(s->h_small);
	for (int b = 0; b < 2; b++) for (int k = 0; k < 3; k++) if (s->ev[b][k]) (void)hipEventDestroy(s->ev[b][k]);
	delete s;
}

tnsx_status tnsx_slab_set_active_search(tnsx_slab* s, int set_i, int set_j, int active)
{
	if (!s || set_i < 0 || set_j < 0) return TNSX_ERR_INVALID;
	for (auto& a : s->active) if (a.first.first == set_i && a.first.second == set_j) { a.second = active ? 1 : 0; s->active_applied = false; return TNSX_OK; }
	s->active.push_back({ { set_i, set_j }, active ? 1 : 0 });
	s->active_applied = false;
	return TNSX_OK;
}

int tnsx_slab_engine_set(const tnsx_slab* s, int set_index)
{
	return (s && set_index >= 0 && (size_t)set_index < s->sets.size()) ? s->sets[(size_t)set_index].set_id : -1;
}

tnsx_status tnsx_slab_get_info(const tnsx_slab* s, tnsx_slab_info* out)
{
	if (!s || !out) return TNSX_ERR_INVALID;
	*out = s->info;
	return TNSX_OK;
}

tnsx_status tnsx_slab_step(tnsx_slab* s, int n_sets, const float* const* xyz, const long long* const* gids, const float* const* radii, const int* n_points)
{
	if (!s || n_sets < 1 || !xyz || !gids || !n_points) return TNSX_ERR_INVALID;
	if (hipSetDevice(s->device) != hipSuccess) return sfail(s, TNSX_ERR_HIP, "hipSetDevice failed");
	for (int k = 0; k < n_sets; k++) {
		if (n_points[k] < 0 || (n_points[k] > 0 && (!xyz[k] || !gids[k]))) return sfail(s, TNSX_ERR_INVALID, "tnsx_slab_step: null pointer or negative size (set %d)", k);
		if (s->variable && n_points[k] > 0 && (!radii || !radii[k])) return sfail(s, TNSX_ERR_INVALID, "per-point radii must be given for every set, or for none (fixed radius)");
	}
	// (the engine's sets, their active pairs and the agreed capacities all hang on the set list of the first step)
	if (!s->sets.empty() && (size_t)n_sets != s->sets.size())
		return sfail(s, TNSX_ERR_INVALID, "tnsx_slab_step: %d sets, but the slab was first stepped with %zu (a set may be empty, n_points = 0, but must be passed)", n_sets, s->sets.size());
	if ((size_t)n_sets > s->sets.size()) s->sets.resize((size_t)n_sets);
	{ const tnsx_status r = ensure_small(s, (size_t)n_sets); if (r != TNSX_OK) return r; }
	const size_t W = s->variable ? 6 : 5;
	const bool side_on[2] = { has_side(s, 0), has_side(s, 1) };
	s->info.rounds_last = 0;
	s->info.redone_last = 0;
	s->ev_rounds = 0;
	s->info.exchange_ms_last = 0.0f;
	SHIP(s, hipMemsetAsync(s->d_small.p, 0, (size_t)n_sets * 8 * 4, s->stream));
	bool spec = s->speculative && (side_on[0] || side_on[1]);
	for (int k = 0; k < n_sets && spec; k++) for (int side = 0; side < 2; side++) if (side_on[side] && !s->sets[(size_t)k].caps_known[side]) spec = false;
	s->info.speculative_last = spec ? 1 : 0;
	auto rad = [&](int k) { return s->variable ? radii[k] : nullptr; };

	if (spec) {
		// ---- ONE round of fixed-capacity messages; nothing is read on the host before the search
		std::vector<tnsx_slab_op> ops;
		for (int k = 0; k < n_sets; k++) {
			SetState& st = s->sets[(size_t)k];
			{ const tnsx_status r = pack(s, (size_t)k, xyz[k], gids[k], rad(k), n_points[k], side_on, st.cap_s, false); if (r != TNSX_OK) return r; }
			for (int side = 0; side < 2; side++) {
				if (!side_on[side]) continue;
				if (!st.recv[side].reserve(((size_t)st.cap_r[side] + 1) * W * 4)) return sfail(s, TNSX_ERR_HIP, "out of device memory (halo buffers)");
				ops.push_back({ peer_of(s, side), st.send[side].p, ((size_t)st.cap_s[side] + 1) * W * 4, st.recv[side].p, ((size_t)st.cap_r[side] + 1) * W * 4 });
			}
		}
		if (const int xrc = do_exchange(s, ops)) return sfail(s, xrc == 3 ? TNSX_ERR_TIMEOUT : TNSX_ERR_HIP, "halo exchange failed (transport)");
		for (int k = 0; k < n_sets; k++) {
			SetState& st = s->sets[(size_t)k];
			const uint32_t rows[2] = { side_on[0] ? st.cap_r[0] : 0u, side_on[1] ? st.cap_r[1] : 0u };
			{ const tnsx_status r = assemble(s, (size_t)k, xyz[k], gids[k], rad(k), n_points[k], rows, true); if (r != TNSX_OK) return r; }
			// what validate() needs, fetched behind the search: the received header words next to the pack counts
			for (int side = 0; side < 2; side++) if (side_on[side]) SHIP(s, hipMemcpyAsync(small_dev(s, (size_t)k) + 2 + side, st.recv[side].p, 4, hipMemcpyDeviceToDevice, s->stream));
		}
		SHIP(s, hipMemcpyAsync(s->h_small, s->d_small.p, (size_t)n_sets * 8 * 4, hipMemcpyDeviceToHost, s->stream));
		{ const tnsx_status r = run_engine(s); if (r != TNSX_OK) return r; }       // (synchronises the stream)
		SWAIT(s, "reading the counts of the speculative exchange");
		// ---- validate: every link on its own.  Both ends of a link see the same two numbers (what travelled, what was agreed).
		bool repair_link[2] = { false, false };
		for (int k = 0; k < n_sets; k++) {
			SetState& st = s->sets[(size_t)k];
			for (int side = 0; side < 2; side++) {
				if (!side_on[side]) continue;
				st.n_out[side] = small_host(s, (size_t)k)[side];
				st.n_in[side] = small_host(s, (size_t)k)[2 + side];
				if (st.n_out[side] > st.cap_s[side] || st.n_in[side] > st.cap_r[side]) repair_link[side] = true;
			}
		}
		if (repair_link[0] || repair_link[1]) {
			// a capacity was exceeded: some ghosts are missing.  The two ends of that link move exactly the rows that did not fit
			// (nobody else takes part), then this slab searches again.
			s->info.redone_last = 1;
			std::vector<tnsx_slab_op> fix;
			for (int k = 0; k < n_sets; k++) {
				SetState& st = s->sets[(size_t)k];
				bool want[2] = { false, false };
				uint32_t cap_rows[2] = { st.cap_s[0], st.cap_s[1] };
				for (int side = 0; side < 2; side++) if (side_on[side] && repair_link[side] && st.n_out[side] > st.cap_s[side]) { want[side] = true; cap_rows[side] = st.n_out[side]; }
				if (want[0] || want[1]) { const tnsx_status r = pack(s, (size_t)k, xyz[k], gids[k], rad(k), n_points[k], want, cap_rows, false); if (r != TNSX_OK) return r; }
				for (int side = 0; side < 2; side++) {
					if (!side_on[side] || !repair_link[side]) continue;
					const bool out_over = st.n_out[side] > st.cap_s[side], in_over = st.n_in[side] > st.cap_r[side];
					if (in_over && !st.recv[side].reserve(((size_t)st.n_in[side] + 1) * W * 4, true)) return sfail(s, TNSX_ERR_HIP, "out of device memory (halo buffers)");
					if (out_over || in_over)
						fix.push_back({ peer_of(s, side), out_over ? (const void*)(st.send[side].as<float>() + W) : nullptr, out_over ? (size_t)st.n_out[side] * W * 4 : 0,
						                in_over ? (void*)(st.recv[side].as<float>() + W) : nullptr, in_over ? (size_t)st.n_in[side] * W * 4 : 0 });
				}
			}
			if (const int xrc = do_exchange(s, fix)) return sfail(s, xrc == 3 ? TNSX_ERR_TIMEOUT : TNSX_ERR_HIP, "halo exchange failed (transport, repair round)");
			SHIP(s, hipMemsetAsync(s->d_small.p, 0, (size_t)n_sets * 8 * 4, s->stream));
			for (int k = 0; k < n_sets; k++) {
				SetState& st = s->sets[(size_t)k];
				const uint32_t rows[2] = { side_on[0] ? st.n_in[0] : 0u, side_on[1] ? st.n_in[1] : 0u };
				{ const tnsx_status r = assemble(s, (size_t)k, xyz[k], gids[k], rad(k), n_points[k], rows, false); if (r != TNSX_OK) return r; }
			}
			SHIP(s, hipMemcpyAsync(s->h_small, s->d_small.p, (size_t)n_sets * 8 * 4, hipMemcpyDeviceToHost, s->stream));
			{ const tnsx_status r = run_engine(s); if (r != TNSX_OK) return r; }
			SWAIT(s, "finishing the repaired step");
		}
	}
	else {
		// ---- exact step: the counts first, then exactly the rows
		std::vector<tnsx_slab_op> ops;
		for (int k = 0; k < n_sets; k++) {
			SetState& st = s->sets[(size_t)k];
			uint32_t cap_rows[2];
			for (int side = 0; side < 2; side++) cap_rows[side] = std::max<uint32_t>({ st.cap_s[side], (uint32_t)(n_points[k] / 32), 1024u });
			for (;;) {
				{ const tnsx_status r = pack(s, (size_t)k, xyz[k], gids[k], rad(k), n_points[k], side_on, cap_rows, true); if (r != TNSX_OK) return r; }
				bool grown = false;
				for (int side = 0; side < 2; side++) if (side_on[side] && st.n_out[side] > cap_rows[side]) { cap_rows[side] = st.n_out[side] + st.n_out[side] / 8 + 1024; grown = true; }
				if (!grown) break;
			}
			for (int side = 0; side < 2; side++) {
				if (!side_on[side]) continue;
				if (!st.recv[side].reserve(std::max<size_t>(((size_t)st.cap_r[side] + 1) * W * 4, W * 4))) return sfail(s, TNSX_ERR_HIP, "out of device memory (halo buffers)");
				ops.push_back({ peer_of(s, side), st.send[side].p, 4, st.recv[side].p, 4 });
			}
		}
		if (const int xrc = do_exchange(s, ops)) return sfail(s, xrc == 3 ? TNSX_ERR_TIMEOUT : TNSX_ERR_HIP, "halo exchange failed (transport, counts)");
		for (int k = 0; k < n_sets; k++) for (int side = 0; side < 2; side++) if (side_on[side])
			SHIP(s, hipMemcpyAsync(small_dev(s, (size_t)k) + 2 + side, s->sets[(size_t)k].recv[side].p, 4, hipMemcpyDeviceToDevice, s->stream));
		SHIP(s, hipMemcpyAsync(s->h_small, s->d_small.p, (size_t)n_sets * 8 * 4, hipMemcpyDeviceToHost, s->stream));
		SWAIT(s, "exchanging the row counts with the neighbours");
		ops.clear();
		for (int k = 0; k < n_sets; k++) {
			SetState& st = s->sets[(size_t)k];
			for (int side = 0; side < 2; side++) {
				if (!side_on[side]) continue;
				st.n_in[side] = small_host(s, (size_t)k)[2 + side];
				if (!st.recv[side].reserve(((size_t)st.n_in[side] + 1) * W * 4)) return sfail(s, TNSX_ERR_HIP, "out of device memory (halo buffers)");
				if (st.n_out[side] || st.n_in[side])
					ops.push_back({ peer_of(s, side), st.n_out[side] ? (const void*)(st.send[side].as<float>() + W) : nullptr, (size_t)st.n_out[side] * W * 4,
					                st.n_in[side] ? (void*)(st.recv[side].as<float>() + W) : nullptr, (size_t)st.n_in[side] * W * 4 });
			}
		}
		if (const int xrc = do_exchange(s, ops)) return sfail(s, xrc == 3 ? TNSX_ERR_TIMEOUT : TNSX_ERR_HIP, "halo exchange failed (transport, rows)");
		for (int k = 0; k < n_sets; k++) {
			SetState& st = s->sets[(size_t)k];
			const uint32_t rows[2] = { side_on[0] ? st.n_in[0] : 0u, side_on[1] ? st.n_in[1] : 0u };
			{ const tnsx_status r = assemble(s, (size_t)k, xyz[k], gids[k], rad(k), n_points[k], rows, false); if (r != TNSX_OK) return r; }
		}
		SHIP(s, hipMemcpyAsync(s->h_small, s->d_small.p, (size_t)n_sets * 8 * 4, hipMemcpyDeviceToHost, s->stream));
		{ const tnsx_status r = run_engine(s); if (r != TNSX_OK) return r; }
		SWAIT(s, "finishing the exact step");
	}
	// ---- capacities for the next step: grow only, the same rule on the same numbers at both ends of a link
	for (int k = 0; k < n_sets; k++) {
		SetState& st = s->sets[(size_t)k];
		for (int side = 0; side < 2; side++) {
			if (!side_on[side]) continue;
			if (st.n_out[side] > st.cap_s[side] || !st.caps_known[side]) st.cap_s[side] = std::max(st.cap_s[side], capacity_rule(st.n_out[side]));
			if (st.n_in[side] > st.cap_r[side] || !st.caps_known[side]) st.cap_r[side] = std::max(st.cap_r[side], capacity_rule(st.n_in[side]));
			st.caps_known[side] = true;
		}
		if (small_host(s, (size_t)k)[4]) return sfail(s, TNSX_ERR_INVALID, "a search radius exceeds max_radius = %g: the halo is too thin for exact results", (double)s->max_radius);
		if (small_host(s, (size_t)k)[5]) return sfail(s, TNSX_ERR_LIST_TOO_LONG, "a global id does not fit the 32-bit neighbour indices of the reference's list layout");
	}
	s->info.n_owned = s->sets[0].n_owned; s->info.n_ghost = s->sets[0].n_ghost;
	// (every path above ends with a wait for the stream: the events of this step's exchange rounds have completed)
	for (int k = 0; k < s->ev_rounds; k++) {
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, s->ev[0][k], s->ev[1][k]) == hipSuccess) s->info.exchange_ms_last += ms;
	}
	return TNSX_OK;
}

tnsx_status tnsx_slab_set_collect_times(tnsx_slab* s, int on)
{
	if (!s) return TNSX_ERR_INVALID;
	s->collect_times = on != 0;
	return TNSX_OK;
}

tnsx_status tnsx_slab_transport_check(tnsx_context* engine, const tnsx_slab_transport* transport, int rank, int world, int* ranks_seen)
{
	if (!engine || !ranks_seen || world < 1 || rank < 0 || rank >= world) return TNSX_ERR_INVALID;
	*ranks_seen = 1;
	if (!transport || !transport->allreduce) return world == 1 ? TNSX_OK : TNSX_ERR_INVALID;
	hipStream_t st = static_cast<hipStream_t>(tnsx_internal_stream(engine));
	if (!st) return TNSX_ERR_STATE;
	if (hipSetDevice(tnsx_internal_device(engine)) != hipSuccess) return TNSX_ERR_HIP;
	uint32_t* d = nullptr;
	if (hipMalloc(&d, sizeof(uint32_t)) != hipSuccess) return TNSX_ERR_HIP;
	const uint32_t one = 1u;
	uint32_t sum = 0u;
	tnsx_status rc = TNSX_OK;
	if (hipMemcpyAsync(d, &one, sizeof one, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = TNSX_ERR_HIP;
	if (rc == TNSX_OK) {
		const int arc = transport->allreduce(transport->user, rank, world, d, 1, TNSX_SLAB_SUM_U32, st);
		if (arc != 0) rc = arc == 3 ? TNSX_ERR_TIMEOUT : TNSX_ERR_HIP;      // (3: the transport's own bounded wait expired)
	}
	if (rc == TNSX_OK && (wait_stream(st, 60.0) != 0 || hipMemcpy(&sum, d, sizeof sum, hipMemcpyDeviceToHost) != hipSuccess)) rc = TNSX_ERR_TIMEOUT;
	(void)hipFree(d);
	if (rc == TNSX_OK) *ranks_seen = (int)sum;
	return rc;
}

tnsx_status tnsx_slab_set_watchdog(tnsx_slab* s, double seconds)
{
	if (!s) return TNSX_ERR_INVALID;
	s->watchdog_s = seconds;
	// (the in-process transport waits for its peers on the host: the same bound)
	if (s->tr.exchange == local_exchange && s->tr.user) {
		LocalGroup* g = static_cast<LocalTransport*>(s->tr.user)->g;
		std::lock_guard<std::mutex> lk(g->mu);      // (peers read it under the lock)
		g->host_wait_s = seconds > 0.0 ? seconds : 1.0e9;
	}
	return TNSX_OK;
}

// ------------------------------------------------------------------------------------------------ redistribution
}  // extern "C"
struct tnsx_slab_redist {
	DBuf rows;            // the rows this rank owns after the exchange, W floats each
	size_t n_rows = 0;
	int W = 5;
	hipStream_t stream = nullptr;
	int device = 0;
};
static double g_redistribute_watchdog_s = 120.0;
extern "C" {

tnsx_status tnsx_slab_set_redistribute_watchdog(double seconds)
{
	g_redistribute_watchdog_s = seconds;   // (<= 0: wait for ever)
	return TNSX_OK;
}

tnsx_status tnsx_slab_redistribute_begin(tnsx_context* engine, const tnsx_slab_transport* tr, int rank, int world, const float* cuts, const float* xyz,
                                         const long long* gids, const float* radii, int n_points, tnsx_slab_redist** out, int* n_owned)
{
	auto fail = [&](tnsx_status st, const char* msg) { g_slab_create_error = std::string("tnsx_slab_redistribute_begin: ") + msg; return st; };
	if (!engine || !cuts || !out || !n_owned || world < 1 || world > 64 || rank < 0 || rank >= world || n_points < 0) return fail(TNSX_ERR_INVALID, "bad argument (1 <= world <= 64)");
	if (n_points > 0 && (!xyz || !gids)) return fail(TNSX_ERR_INVALID, "null point or id array");
	if (world > 1 && (!tr || !tr->exchange)) return fail(TNSX_ERR_INVALID, "world > 1 needs a transport");
	for (int k = 1; k < world; k++) if (!(cuts[k] >= cuts[k - 1])) return fail(TNSX_ERR_INVALID, "cuts must ascend");
	const int device = tnsx_internal_device(engine);
	hipStream_t stream = static_cast<hipStream_t>(tnsx_internal_stream(engine));
	if (!stream) return fail(TNSX_ERR_STATE, "multi-device contexts shard host data themselves");
	if (hipSetDevice(device) != hipSuccess) return fail(TNSX_ERR_HIP, "hipSetDevice failed");
	const int W = radii ? 6 : 5;
	const double watchdog = g_redistribute_watchdog_s;   // (tnsx_slab_set_redistribute_watchdog; default 120 s)
	// ---- how many of my points go where
	DBuf d_small;   // [0, world) send counts = scatter cursors, [world, 2 world) first row of every destination, [2 world, 3 world) receive counts
	if (!d_small.reserve((size_t)3 * world * 4)) return fail(TNSX_ERR_HIP, "out of device memory");
	unsigned int* d_cnt = d_small.as<unsigned int>(), *d_first = d_cnt + world, *d_rcnt = d_first + world;
	if (hipMemsetAsync(d_small.p, 0, (size_t)3 * world * 4, stream) != hipSuccess) return fail(TNSX_ERR_HIP, "memset failed");
	tnsx::launch_slab_dest_rows(true, xyz, radii, gids, n_points, cuts, world, d_cnt, nullptr, nullptr, W, stream);
	std::vector<unsigned int> h_cnt((size_t)world, 0u), h_rcnt((size_t)world, 0u), h_first((size_t)world, 0u);
	// ---- round 1: the counts, to and from every other rank
	std::vector<tnsx_slab_op> ops;
	for (int p = 0; p < world; p++) if (p != rank) ops.push_back({ p, d_cnt + p, 4, d_rcnt + p, 4 });
	if (!ops.empty() && tr->exchange(tr->user, rank, world, ops.data(), (int)ops.size(), stream)) return fail(TNSX_ERR_HIP, "exchange of the counts failed (transport)");
	if (hipMemcpyAsync(h_cnt.data(), d_cnt, (size_t)world * 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
	    hipMemcpyAsync(h_rcnt.data(), d_rcnt, (size_t)world * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return fail(TNSX_ERR_HIP, "copy of the counts failed");
	{
		const int w = wait_stream(stream, watchdog);
		if (w == 2) {
			// (with an abort hook the pending operations return and the stream drains; without one -- in-process, host-staged, an application's own transport -- the
			//  stream may still be stuck, or complete later: the buffers it may touch are leaked, not freed)
			if (tr && tr->abort) tr->abort(tr->user); else d_small.abandon();
			return fail(TNSX_ERR_TIMEOUT, "the exchange of the counts did not complete within the watchdog's time (a rank that never called, or a transport that does not reach every pair of ranks)");
		}
		if (w == 1) return fail(TNSX_ERR_HIP, "HIP error while exchanging the counts");
	}
	h_rcnt[(size_t)rank] = h_cnt[(size_t)rank];   // my own share stays
	size_t n_send = 0, n_recv = 0;
	for (int p = 0; p < world; p++) { h_first[(size_t)p] = (unsigned int)n_send; n_send += h_cnt[(size_t)p]; n_recv += h_rcnt[(size_t)p]; }
	if (n_recv > 0x7fffffffull) return fail(TNSX_ERR_LIST_TOO_LONG, "a slab would own more than 2^31 - 1 points");
	// ---- my points as rows, grouped by destination
	DBuf send;
	tnsx_slab_redist* r = new tnsx_slab_redist();
	r->W = W; r->stream = stream; r->device = device; r->n_rows = n_recv;
	if (!send.reserve(std::max<size_t>(n_send, 1) * W * 4) || !r->rows.reserve(std::max<size_t>(n_recv, 1) * W * 4)) { delete r; return fail(TNSX_ERR_HIP, "out of device memory (rows)"); }
	if (hipMemcpyAsync(d_first, h_first.data(), (size_t)world * 4, hipMemcpyHostToDevice, stream) != hipSuccess ||
	    hipMemsetAsync(d_cnt, 0, (size_t)world * 4, stream) != hipSuccess) { delete r; return fail(TNSX_ERR_HIP, "copy failed"); }
	tnsx::launch_slab_dest_rows(false, xyz, radii, gids, n_points, cuts, world, d_cnt, d_first, send.as<float>(), W, stream);
	// ---- round 2: the rows.  Receive layout: the shares of rank 0, 1, ... one behind the other.
	ops.clear();
	size_t roff = 0;
	for (int p = 0; p < world; p++) {
		const size_t sb = (size_t)h_cnt[(size_t)p] * W * 4, rb = (size_t)h_rcnt[(size_t)p] * W * 4;
		const float* src = send.as<float>() + (size_t)h_first[(size_t)p] * W;
		float* dst = r->rows.as<float>() + roff * W;
		if (p == rank) { if (sb && hipMemcpyAsync(dst, src, sb, hipMemcpyDeviceToDevice, stream) != hipSuccess) { delete r; return fail(TNSX_ERR_HIP, "copy failed"); } }
		else if (sb || rb) ops.push_back({ p, sb ? (const void*)src : nullptr, sb, rb ? (void*)dst : nullptr, rb });
		roff += h_rcnt[(size_t)p];
	}
	if (!ops.empty() && tr->exchange(tr->user, rank, world, ops.data(), (int)ops.size(), stream)) { delete r; return fail(TNSX_ERR_HIP, "exchange of the rows failed (transport)"); }
	{
		const int w = wait_stream(stream, watchdog);   // (the send buffer is released below)
		if (w == 2) {
			if (tr && tr->abort) { tr->abort(tr->user); delete r; }
			else { d_small.abandon(); send.abandon(); r->rows.abandon(); delete r; }   // (see above: leaked on purpose)
			return fail(TNSX_ERR_TIMEOUT, "the exchange of the rows did not complete within the watchdog's time");
		}
		if (w == 1) { delete r; return fail(TNSX_ERR_HIP, "HIP error while exchanging the rows"); }
	}
	*out = r;
	*n_owned = (int)n_recv;
	return TNSX_OK;
}

tnsx_status tnsx_slab_redistribute_finish(tnsx_slab_redist* r, float* xyz_out, long long* gids_out, float* radii_out)
{
	if (!r) return TNSX_ERR_INVALID;
	tnsx_status st = TNSX_OK;
	if (xyz_out || gids_out) {
		if (!xyz_out || !gids_out || (r->W == 6 && !radii_out)) { g_slab_create_error = "tnsx_slab_redistribute_finish: xyz, ids (and radii, if radii were given) are needed together"; st = TNSX_ERR_INVALID; }
		else if (hipSetDevice(r->device) != hipSuccess) st = TNSX_ERR_HIP;
		else {
			tnsx::launch_slab_rows_to_points(r->rows.as<float>(), r->n_rows, r->W, xyz_out, r->W == 6 ? radii_out : nullptr, gids_out, r->stream);
			if (hipStreamSynchronize(r->stream) != hipSuccess) st = TNSX_ERR_HIP;   // (the rows are freed with the handle)
		}
	}
	delete r;
	return st;
}

tnsx_status tnsx_slab_debug_set_capacity(tnsx_slab* s, int side, unsigned rows)
{
	// tests: pretend the halos were thinner when the capacities were agreed (both ends of the link must be given the same number)
	if (!s || side < 0 || side > 1) return TNSX_ERR_INVALID;
	for (SetState& st : s->sets) if (st.caps_known[side]) { st.cap_s[side] = rows; st.cap_r[side] = rows; }
	return TNSX_OK;
}

}  // extern "C"
