#include "kdtree.hpp"

#include <algorithm>
#include <thread>

namespace fy {
namespace {

struct Elem { double x[3]; int32_t id; };

struct CmpAxis {   // meshTree.H:45-55 cmpvec: strict < on one coordinate
    int a;
    bool operator()(const Elem& p, const Elem& q) const { return p.x[a] < q.x[a]; }
};

// meshTree.C:19-37.  The subtree over pts[lo,hi) occupies nodes[o, o+(hi-lo)).
void build_rec(Elem* pts, int64_t lo, int64_t hi, int depth, KdNode* nodes, int64_t o, int par_levels) {
    while (hi > lo) {
        const int axis = depth % 3;
        const int64_t n = hi - lo, md = lo + n / 2;
        std::nth_element(pts + lo, pts + md, pts + hi, CmpAxis{axis});
        KdNode& nd = nodes[o];
        nd.x = pts[md].x[0]; nd.y = pts[md].x[1]; nd.z = pts[md].x[2]; nd.id = pts[md].id; nd.pad = 0;
        const int64_t nl = n / 2;
        if (par_levels > 0 && n > 4096) {
            // the two subtrees touch disjoint element and node ranges
            std::thread t(build_rec, pts, lo, md, depth + 1, nodes, o + 1, par_levels - 1);
            build_rec(pts, md + 1, hi, depth + 1, nodes, o + 1 + nl, par_levels - 1);
            t.join();
            return;
        }
        build_rec(pts, lo, md, depth + 1, nodes, o + 1, 0);
        // tail-iterate on the right subtree
        lo = md + 1; o = o + 1 + nl; depth += 1;
    }
}

}  // namespace

void build_kdtree_preorder(const double* centres, int32_t n_cells, std::vector<KdNode>& nodes, int threads) {
    std::vector<Elem> pts((size_t)n_cells);
    for (int32_t c = 0; c < n_cells; ++c) {      // meshTree.C:12-14: cell order
        pts[c].x[0] = centres[3 * (size_t)c]; pts[c].x[1] = centres[3 * (size_t)c + 1]; pts[c].x[2] = centres[3 * (size_t)c + 2];
        pts[c].id = c;
    }
    nodes.resize((size_t)n_cells);
    int par_levels = 0;
    while ((1 << par_levels) < threads) ++par_levels;
    build_rec(pts.data(), 0, n_cells, 0, nodes.data(), 0, par_levels);
}

int kdtree_levels(int64_t n) {
    int l = 0;
    while (n > 0) { ++l; n = n / 2; }   // the left child (n/2 nodes) is never smaller than the right one
    return l;
}

}  // namespace fy
