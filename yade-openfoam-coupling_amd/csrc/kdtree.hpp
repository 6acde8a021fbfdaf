// k-d tree over cell centres, flattened.  Replaces Foam::meshTree's pointer graph (meshTree.H:16-33: three
// pointers + heap vector per node, leaked `new kdNode` per cell) with one preorder array of 32-byte nodes.
//
// The tree SHAPE depends only on the element count (node = element n/2 after nth_element, left = first n/2
// elements, right = the rest; meshTree.C:27-31), so no child pointers are stored: the node at preorder offset o
// with subtree size n has  left = (o+1, n/2)  and  right = (o+1+n/2, n-n/2-1).  WHICH cell sits at a node is the
// result of std::nth_element's tie-breaking on a lattice full of equal coordinates (meshTree.C:50); we run the
// same library algorithm on the same element sequence in place (host, once per mesh -- it is construction-time
// work in the reference too, FoamYade.C:33) and tests pin the result against the reference's own tree.
#pragma once
#include <cstdint>
#include <vector>

namespace fy {

struct alignas(32) KdNode {
    double x, y, z;   // cell centre (bit copy of mesh.C()[id])
    int32_t id;       // cell id
    int32_t pad;
};

// nodes.size() == n_cells on return; `threads` parallelises independent subtrees (result is identical).
void build_kdtree_preorder(const double* centres, int32_t n_cells, std::vector<KdNode>& nodes, int threads);

// number of levels of the implicit tree with n nodes
int kdtree_levels(int64_t n);

}  // namespace fy
