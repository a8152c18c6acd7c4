// Multi-device mode of the C ABI (tnsx_options.n_devices > 1): ONE context that shards every run over several GPUs of the node,
// for callers that hand over HOST pointers -- the C++ drop-in `tns::TreeNSearch` above all.  SURVEY.md section 8(e) applied to a
// single process: slabs along x with balanced cuts, one engine per GPU over [owned | ghosts], candidates-only ghosts, global ids
// emitted by the engines -- but since the points start in host memory the "halo exchange" is free: the host thread that
// uploads slab k simply adds the ghost points to the upload.  What the caller gains is the node's aggregate PCIe bandwidth:
// in drop-in mode a run is bound by copying the lists back (2.5 GB at 10 M points), and N devices copy over N links at once.
//
// Internal interface between tnsx_engine.cpp (which owns the C ABI) and tnsx_multi.cpp.
#pragma once
#include "../../include/tnsx.h"

#include <cstdint>
#include <string>
#include <vector>

namespace tnsx_multi {

struct State;   // opaque

// creates the per-device engines (through the C ABI itself, one single-device context each); nullptr + message on failure
State* create(const tnsx_options& opt, std::string& error);
void destroy(State* m);

// the subset of the ABI that makes sense for host-resident inputs; each returns a tnsx_status and leaves the message in `error`
int add_point_set(State* m, const void* xyz, const void* radii, int n, unsigned flags, std::string& error);
tnsx_status resize_point_set(State* m, int set_id, const void* xyz, const void* radii, int n, unsigned flags, std::string& error);
tnsx_status set_search_radius(State* m, float r, std::string& error);
tnsx_status set_cell_size(State* m, float cell, std::string& error);
void set_symmetric(State* m, bool on);
void set_arithmetic(State* m, int arith);
tnsx_status set_active(State* m, int i, int j, bool on, std::string& error);
tnsx_status set_active_all(State* m, int i, bool search_in_all, bool be_found_by_all, std::string& error);
void set_all_searches(State* m, bool on);
int n_sets(const State* m);
int n_points_in_set(const State* m, int s);
int64_t total_points(const State* m);
bool is_active(const State* m, int i, int j);
uint64_t neighborlist_bytes(const State* m);

tnsx_status run(State* m, std::string& error);
tnsx_status pair_view(State* m, int i, int j, tnsx_csr_view* out, std::string& error);
tnsx_status prepare_zsort(State* m, std::string& error);
tnsx_status zsort_order(State* m, int set_i, const int** host, int* n, std::string& error);
tnsx_status apply_zsort(State* m, int set_i, void* data, size_t elem_bytes, int stride, std::string& error);
void stats(const State* m, tnsx_stats* out);

}  // namespace tnsx_multi
