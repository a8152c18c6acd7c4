// tns::NeighborList -- handle to one neighbour record `[count, j0, j1, ...]` inside the engine's pinned host
// mirror.  Same public surface as the reference handle (TreeNSearch/source/NeighborList.h:8-39): size(),
// operator[], get_ptr(); constructible only by tns::TreeNSearch.
#pragma once
#include <cstddef>

namespace tns
{
	class TreeNSearch;

	class NeighborList
	{
	public:
		/** Number of neighbours in the list. */
		inline int size() const { return record_[0]; }
		/** Index (set-local, into set_j) of the i-th neighbour. */
		inline int operator[](const size_t i) const { return record_[1 + i]; }
		/** Pointer to the first neighbour index; size() entries are valid. */
		inline const int* get_ptr() const { return record_ + 1; }

	private:
		friend class TreeNSearch;
		explicit NeighborList(const int* record) : record_(record) {}
		const int* record_;   // points at the count word
	};
}
