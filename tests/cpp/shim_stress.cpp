// The two stress scenarios of the reference's test program, driven through the header-only drop-in (`#include <TreeNSearch>`)
// and checked against an all-pairs search written here (the reference checks them against tests/BruteforceNSearch; for the
// size lattice it only looks for crashes -- here every configuration is compared):
//
//   emitter   two variable-radius sets that start EMPTY (null pointers, n = 0) and are resized at random -- grow, shrink,
//             replace -- step after step; all four searches active; every step is compared   (tests/tests.cpp:434-514)
//   lattice   1..3 sets, every combination of set sizes from a list with the awkward ones (0, 1, 2, ... 9, 15, 16, 17, 63, 64,
//             65, 100, 1000), random coordinates and per-point radii, all searches active: run, compare, prepare_zsort, apply_zsort
//             to coordinates and radii, run, compare                                           (tests/tests.cpp:287-427)
//
// usage: shim_stress [emitter_steps] [lattice: 0 = skip, 1 = reduced size lists for 2 and 3 sets (default), 2 = full list everywhere]
// Compiled with -ffp-contract=off: the all-pairs distance is the STRICT arithmetic the engine defaults to.
#include <TreeNSearch>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>

#include <omp.h>

namespace {

struct SetData {
	std::vector<float> xyz, r;
	int n() const { return (int)r.size(); }
};

std::vector<std::vector<int>> all_pairs(const SetData& a, const SetData& b, bool same, bool symmetric)
{
	std::vector<std::vector<int>> out((size_t)a.n());
	#pragma omp parallel for schedule(dynamic, 16) if (a.n() * (long long)b.n() > 200000)
	for (int i = 0; i < a.n(); i++) {
		const float r2i = a.r[(size_t)i] * a.r[(size_t)i];
		for (int j = 0; j < b.n(); j++) {
			if (same && i == j) continue;
			const float dx = a.xyz[3 * (size_t)i] - b.xyz[3 * (size_t)j], dy = a.xyz[3 * (size_t)i + 1] - b.xyz[3 * (size_t)j + 1],
			            dz = a.xyz[3 * (size_t)i + 2] - b.xyz[3 * (size_t)j + 2];
			const float d2 = (dx * dx + dy * dy) + dz * dz;
			const float r2j = b.r[(size_t)j] * b.r[(size_t)j];
			if (d2 <= r2i || (symmetric && d2 <= r2j)) out[(size_t)i].push_back(j);
		}
	}
	return out;
}

long long g_lists = 0;

bool same_lists(tns::TreeNSearch& ns, const std::vector<SetData>& sets)
{
	const int n_sets = (int)sets.size();
	for (int i = 0; i < n_sets; i++) {
		for (int j = 0; j < n_sets; j++) {
			const std::vector<std::vector<int>> ref = all_pairs(sets[(size_t)i], sets[(size_t)j], i == j, true);
			int bad = 0;
			#pragma omp parallel for schedule(static) reduction(+ : bad) if (sets[(size_t)i].n() > 2000)
			for (int p = 0; p < sets[(size_t)i].n(); p++) {
				const tns::NeighborList nl = ns.get_neighborlist(i, j, p);
				std::vector<int> got(nl.get_ptr(), nl.get_ptr() + nl.size());
				std::sort(got.begin(), got.end());
				if (got != ref[(size_t)p]) bad++;
			}
			g_lists += sets[(size_t)i].n();
			if (bad) { std::printf("\tpair %d->%d: %d lists differ from the all-pairs search\n", i, j, bad); return false; }
		}
	}
	return true;
}

float* ptr(std::vector<float>& v) { return v.empty() ? nullptr : v.data(); }

int emitter(int steps)
{
	std::printf("dynamic emitter, %d steps\n", steps);
	const int n_sets = 2;
	tns::TreeNSearch ns;
	ns.set_n_threads(8);
	std::vector<SetData> sets((size_t)n_sets);
	for (int s = 0; s < n_sets; s++) ns.add_point_set((float*)nullptr, (float*)nullptr, 0);
	ns.set_all_searches(true);
	std::mt19937 gen(123);
	std::uniform_real_distribution<float> coord(0.0f, 10.0f);
	std::uniform_int_distribution<int> pick_set(0, n_sets - 1), pick_action(0, 2), pick_amount(1, 20);
	for (int it = 0; it < steps; it++) {
		const int s = pick_set(gen), action = pick_action(gen), amount = pick_amount(gen);
		SetData& d = sets[(size_t)s];
		int n_new = d.n();
		if (action == 0) n_new += amount;                       // emit
		else if (action == 1) n_new = std::max(0, n_new - amount);   // delete
		else n_new = amount;                                    // replace
		d.xyz.resize(3 * (size_t)n_new);
		d.r.assign((size_t)n_new, 0.5f);
		for (float& v : d.xyz) v = coord(gen);
		ns.resize_point_set(s, ptr(d.xyz), ptr(d.r), n_new);
		ns.run();
		if (!same_lists(ns, sets)) { std::printf("emitter FAILED at step %d (set %d -> %d points)\n", it, s, n_new); return 1; }
	}
	std::printf("emitter passed (%lld lists compared)\n", g_lists);
	return 0;
}

int lattice(int level)
{
	const std::vector<int> full = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 63, 64, 65, 100, 1000 };
	const std::vector<int> reduced2 = { 0, 1, 2, 7, 8, 9, 16, 17, 64, 65, 1000 };   // level 1: 121 pairs, 216 triples (a context per configuration
	const std::vector<int> reduced3 = { 0, 1, 9, 64, 65, 1000 };                    // costs a stream, pinned buffers and a cold first run)
	std::mt19937 gen(42);
	std::uniform_real_distribution<float> coord(0.0f, 10.0f);
	int n_cases = 0;
	for (int n_sets = 1; n_sets <= 3; n_sets++) {
		std::vector<int> sizes = level >= 2 || n_sets == 1 ? full : (n_sets == 2 ? reduced2 : reduced3);
		if (n_sets == 1) for (int k = 0; k < 10; k++) sizes.push_back(10000 + k);
		std::vector<std::vector<int>> combos;
		std::vector<int> cur((size_t)n_sets);
		std::function<void(int)> rec = [&](int d) {
			if (d == n_sets) { combos.push_back(cur); return; }
			for (int c : sizes) { cur[(size_t)d] = c; rec(d + 1); }
		};
		rec(0);
		std::printf("size lattice, %d set(s): %zu combinations\n", n_sets, combos.size());
		for (const std::vector<int>& counts : combos) {
			tns::TreeNSearch ns;
			std::vector<SetData> sets((size_t)n_sets);
			for (int s = 0; s < n_sets; s++) {
				SetData& d = sets[(size_t)s];
				d.xyz.resize(3 * (size_t)counts[(size_t)s]);
				d.r.resize((size_t)counts[(size_t)s]);
				for (int i = 0; i < counts[(size_t)s]; i++) {
					for (int k = 0; k < 3; k++) d.xyz[3 * (size_t)i + (size_t)k] = coord(gen);
					d.r[(size_t)i] = 0.5f + 0.5f * coord(gen) / 10.0f;
				}
				ns.add_point_set(ptr(d.xyz), ptr(d.r), counts[(size_t)s]);
			}
			ns.set_all_searches(true);
			ns.run();
			bool ok = same_lists(ns, sets);
			if (ok) {
				ns.prepare_zsort();
				for (int s = 0; s < n_sets; s++) {
					if (counts[(size_t)s] > 0) {
						ns.apply_zsort(s, sets[(size_t)s].xyz.data(), 3);
						ns.apply_zsort(s, sets[(size_t)s].r.data(), 1);
					}
				}
				ns.run();
				ok = same_lists(ns, sets);
			}
			if (!ok) {
				std::printf("size lattice FAILED for counts [");
				for (int c : counts) std::printf(" %d", c);
				std::printf(" ]\n");
				return 1;
			}
			n_cases++;
		}
	}
	std::printf("size lattice passed (%d configurations, %lld lists compared so far)\n", n_cases, g_lists);
	return 0;
}

}  // namespace

int main(int argc, char** argv)
{
	const int steps = argc > 1 ? std::atoi(argv[1]) : 400;
	const int level = argc > 2 ? std::atoi(argv[2]) : 1;
	// a few threads are plenty for sets of at most 10 k points (a 256-thread team spinning between thousands of tiny parallel
	// regions is what made the first version of this driver take half an hour)
	omp_set_num_threads(std::min(omp_get_max_threads(), 8));
	if (steps > 0 && emitter(steps)) return 1;
	if (level > 0 && lattice(level)) return 1;
	std::printf("ALL PASSED\n");
	return 0;
}
