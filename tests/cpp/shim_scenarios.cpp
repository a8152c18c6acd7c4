// Drives the header-only drop-in (`#include <TreeNSearch>`) through the scenarios of the reference's test program
// (one set fixed radius, two sets variable radius, mixed float/double, resize, zsort round trip; tests.cpp:34-237)
// and compares every neighbour list with an all-pairs search written here.  Compiled with -ffp-contract=off so the
// all-pairs distance is the STRICT arithmetic the engine defaults to.  Prints "ALL PASSED" on success.
#include <TreeNSearch>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <vector>

#include <omp.h>

namespace {

struct Cloud {
	std::vector<std::array<float, 3>> p;
	float search_radius = 0.f;
	int n() const { return (int)p.size(); }
};

Cloud lattice(float lo, float hi, float spacing)
{
	Cloud c;
	c.search_radius = 1.99f * spacing;
	for (float x = lo; x <= hi; x += spacing)
		for (float y = lo; y <= hi; y += spacing)
			for (float z = lo; z <= hi; z += spacing) c.p.push_back({ x, y, z });
	return c;
}

// all-pairs search of set a in set b
std::vector<std::vector<int>> all_pairs(const float* a, const float* ra, int na, const float* b, const float* rb, int nb, bool same, bool symmetric)
{
	std::vector<std::vector<int>> out((size_t)na);
	#pragma omp parallel for schedule(dynamic, 64)
	for (int i = 0; i < na; i++) {
		const float r2i = ra[i] * ra[i];
		for (int j = 0; j < nb; j++) {
			if (same && i == j) continue;
			const float dx = a[3 * i] - b[3 * j], dy = a[3 * i + 1] - b[3 * j + 1], dz = a[3 * i + 2] - b[3 * j + 2];
			const float d2 = (dx * dx + dy * dy) + dz * dz;
			const float r2j = rb[j] * rb[j];
			if (d2 <= r2i || (symmetric && d2 <= r2j)) out[(size_t)i].push_back(j);
		}
	}
	return out;
}

int g_failures = 0;

void expect_equal(tns::TreeNSearch& ns, int si, int sj, const std::vector<std::vector<int>>& ref, const char* what)
{
	int bad = 0;
	// get_neighborlist is a pure read and is called from many user threads (TreeNSearch.cpp:241-249)
	#pragma omp parallel for schedule(static) reduction(+ : bad)
	for (int i = 0; i < (int)ref.size(); i++) {
		const tns::NeighborList nl = ns.get_neighborlist(si, sj, i);
		std::vector<int> got(nl.get_ptr(), nl.get_ptr() + nl.size());
		std::sort(got.begin(), got.end());
		if (got != ref[(size_t)i]) bad++;
	}
	std::printf("\t%-42s %d->%d ... %s\n", what, si, sj, bad ? "xxxxxxx FAILED! xxxxxxx" : "passed!");
	if (bad) g_failures++;
}

void one_set_fixed_radius(int n_points)
{
	std::printf("One point set. Fixed search radius. (%d)\n", n_points);
	Cloud c = lattice(-1.f, 1.f, (float)(2.0 / std::pow((double)n_points, 1.0 / 3.0)));
	std::vector<float> r((size_t)c.n(), c.search_radius);
	tns::TreeNSearch ns;
	ns.set_search_radius(c.search_radius);
	const int s = ns.add_point_set(c.p[0].data(), c.n());
	ns.set_active_search(s, s, true);
	ns.run();
	expect_equal(ns, s, s, all_pairs(c.p[0].data(), r.data(), c.n(), c.p[0].data(), r.data(), c.n(), true, false), "run");
	ns.run_scalar();
	expect_equal(ns, s, s, all_pairs(c.p[0].data(), r.data(), c.n(), c.p[0].data(), r.data(), c.n(), true, false), "run_scalar");
	// zsort round trip
	ns.prepare_zsort();
	ns.apply_zsort(s, c.p[0].data(), 3);
	ns.run();
	expect_equal(ns, s, s, all_pairs(c.p[0].data(), r.data(), c.n(), c.p[0].data(), r.data(), c.n(), true, false), "zsort + run");
	int visited = 0;
	ns.for_each_neighbor(s, s, 0, [&](int) { visited++; });
	if (visited != ns.get_neighborlist(s, s, 0).size()) { std::printf("\tfor_each_neighbor xxxxxxx FAILED! xxxxxxx\n"); g_failures++; }
	// the class is copyable like the reference's (TreeNSearch.h:36-37): a copy carries the configuration and runs on its own
	{
		tns::TreeNSearch copy(ns);
		copy.run();
		expect_equal(copy, s, s, all_pairs(c.p[0].data(), r.data(), c.n(), c.p[0].data(), r.data(), c.n(), true, false), "copy constructed + run");
		tns::TreeNSearch assigned;
		assigned = copy;
		assigned.run();
		expect_equal(assigned, s, s, all_pairs(c.p[0].data(), r.data(), c.n(), c.p[0].data(), r.data(), c.n(), true, false), "copy assigned + run");
		expect_equal(ns, s, s, all_pairs(c.p[0].data(), r.data(), c.n(), c.p[0].data(), r.data(), c.n(), true, false), "original after the copies");
	}
}

void two_sets_variable(int n_points, bool second_as_double)
{
	std::printf("Two point sets. Variable search radius.%s (%d)\n", second_as_double ? " Second set double." : "", n_points);
	const float d = (float)(2.0 / std::pow((double)n_points, 1.0 / 3.0));
	Cloud c0 = lattice(-1.f, 1.f, d), c1 = lattice(-1.f, 1.f, (second_as_double ? 1.33f : 1.31f) * d);
	std::vector<float> r0((size_t)c0.n(), c0.search_radius), r1((size_t)c1.n(), c1.search_radius);
	std::vector<double> p1d((size_t)c1.n() * 3), r1d((size_t)c1.n());
	for (int i = 0; i < c1.n(); i++) { for (int k = 0; k < 3; k++) p1d[(size_t)3 * i + k] = c1.p[(size_t)i][(size_t)k]; r1d[(size_t)i] = r1[(size_t)i]; }
	tns::TreeNSearch ns;
	ns.add_point_set(c0.p[0].data(), r0.data(), c0.n());
	if (second_as_double) ns.add_point_set(p1d.data(), r1d.data(), c1.n()); else ns.add_point_set(c1.p[0].data(), r1.data(), c1.n());
	ns.set_active_search(0, 0, true);
	ns.set_active_search(0, 1, true);
	ns.set_active_search(1, 0, true);
	ns.run();
	expect_equal(ns, 0, 0, all_pairs(c0.p[0].data(), r0.data(), c0.n(), c0.p[0].data(), r0.data(), c0.n(), true, true), "run");
	expect_equal(ns, 0, 1, all_pairs(c0.p[0].data(), r0.data(), c0.n(), c1.p[0].data(), r1.data(), c1.n(), false, true), "run");
	expect_equal(ns, 1, 0, all_pairs(c1.p[0].data(), r1.data(), c1.n(), c0.p[0].data(), r0.data(), c0.n(), false, true), "run");
	ns.set_symmetric_search(false);
	ns.run();
	expect_equal(ns, 0, 1, all_pairs(c0.p[0].data(), r0.data(), c0.n(), c1.p[0].data(), r1.data(), c1.n(), false, false), "asymmetric");
	expect_equal(ns, 1, 0, all_pairs(c1.p[0].data(), r1.data(), c1.n(), c0.p[0].data(), r0.data(), c0.n(), false, false), "asymmetric");
}

void resize_variable(int n_points)
{
	std::printf("Two dynamic point sets with resizes. Variable search radius. (%d)\n", n_points);
	const float d = (float)(2.0 / std::pow((double)n_points, 1.0 / 3.0));
	Cloud c0 = lattice(-1.f, 1.f, d), c1 = lattice(-1.f, 1.f, 1.31f * d);
	std::vector<float> r0((size_t)c0.n(), c0.search_radius), r1((size_t)c1.n(), c1.search_radius);
	tns::TreeNSearch ns;
	ns.add_point_set(c0.p[0].data(), r0.data(), c0.n() / 2);
	ns.add_point_set(c1.p[0].data(), r1.data(), c1.n() / 2);
	ns.set_active_search(0, 0, true);
	ns.set_active_search(0, 1, true);
	ns.set_active_search(1, 0, true);
	const int div[3] = { 2, 1, 3 };
	const char* names[3] = { "original", "resize x2", "resize x0.33" };
	for (int step = 0; step < 3; step++) {
		const int n0 = c0.n() / div[step], n1 = c1.n() / div[step];
		if (step > 0) {
			ns.resize_point_set(0, c0.p[0].data(), r0.data(), n0);
			ns.resize_point_set(1, c1.p[0].data(), r1.data(), n1);
		}
		ns.run();
		expect_equal(ns, 0, 0, all_pairs(c0.p[0].data(), r0.data(), n0, c0.p[0].data(), r0.data(), n0, true, true), names[step]);
		expect_equal(ns, 0, 1, all_pairs(c0.p[0].data(), r0.data(), n0, c1.p[0].data(), r1.data(), n1, false, true), names[step]);
		expect_equal(ns, 1, 0, all_pairs(c1.p[0].data(), r1.data(), n1, c0.p[0].data(), r0.data(), n0, false, true), names[step]);
	}
}

}  // namespace

int main()
{
	for (int n : { 1, 100, 10000 }) {
		std::printf("\nTests with %d particles\n======================================\n", n);
		one_set_fixed_radius(n);
		two_sets_variable(n, false);
		two_sets_variable(n, true);
		resize_variable(n);
	}
	if (g_failures) { std::printf("\n%d checks FAILED\n", g_failures); return 1; }
	std::printf("\nALL PASSED\n");
	return 0;
}
