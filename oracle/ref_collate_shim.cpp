// C entry points over the reference's own collate-time C++, for tests only.
// Compiled TOGETHER WITH the reference sources where they lie (oracle/Makefile):
//   Diff-Reg-3dmatch/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp   batch_grid_subsampling
//   Diff-Reg-3dmatch/cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp                    batch_nanoflann_neighbors
//   Diff-Reg-3dmatch/cpp_wrappers/cpp_utils/cloud/cloud.cpp, cpp_utils/nanoflann/nanoflann.hpp
// This file only converts flat arrays to the std::vector arguments those functions take -- what the reference's
// CPython wrappers (cpp_subsampling/wrapper.cpp:300-420, cpp_neighbors/wrapper.cpp:40-200) do with numpy arrays.
#include <cstring>
#include <vector>
#include "cpp_subsampling/grid_subsampling/grid_subsampling.h"
#include "cpp_neighbors/neighbors/neighbors.h"

extern "C" {

// points [n,3] float32, lengths [nb] int32 -> out_points [<= n,3], out_lengths [nb]; returns the number of points written
int ref_subsample_batch(const float* points, int n, const int* lengths, int nb, float sampleDl, int max_p, float* out_points,
                        int* out_lengths) {
    std::vector<PointXYZ> op(n), sp;
    for (int i = 0; i < n; ++i) op[i] = PointXYZ(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    std::vector<float> of, sf;
    std::vector<int> oc, sc, ob(lengths, lengths + nb), sb;
    batch_grid_subsampling(op, sp, of, sf, oc, sc, ob, sb, sampleDl, max_p);
    for (size_t i = 0; i < sp.size(); ++i) { out_points[3 * i] = sp[i].x; out_points[3 * i + 1] = sp[i].y; out_points[3 * i + 2] = sp[i].z; }
    for (int b = 0; b < nb; ++b) out_lengths[b] = sb[b];
    return (int)sp.size();
}

// -> width of the neighbour matrix (max count); out [nq, width] int32 is written when out != NULL and cap >= nq * width
int ref_batch_query(const float* queries, int nq, const float* supports, int ns, const int* q_lengths, const int* s_lengths, int nb,
                    float radius, int* out, long cap) {
    std::vector<PointXYZ> q(nq), s(ns);
    for (int i = 0; i < nq; ++i) q[i] = PointXYZ(queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]);
    for (int i = 0; i < ns; ++i) s[i] = PointXYZ(supports[3 * i], supports[3 * i + 1], supports[3 * i + 2]);
    std::vector<int> qb(q_lengths, q_lengths + nb), sb(s_lengths, s_lengths + nb), nbr;
    batch_nanoflann_neighbors(q, s, qb, sb, nbr, radius);
    const int width = nq > 0 ? (int)(nbr.size() / nq) : 0;
    if (out && cap >= (long)nbr.size()) std::memcpy(out, nbr.data(), nbr.size() * sizeof(int));
    return width;
}

}
