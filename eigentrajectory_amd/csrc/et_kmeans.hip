// et_kmeans.hip -- BatchKMeans (EigenTrajectory/kmeans.py) for gfx950.
//
//   euc_sim            kmeans.py:59-76    y = (2 a.b - |a|^2) - |b|^2, a.b an fmaf chain
//   kmeanspp           kmeans.py:78-112   farthest-first: running max-similarity + global arg-min
//   get_labels         kmeans.py:143-158  arg-max (first max wins, NaN wins)
//   compute_centroids  kmeans.py:160-198  per-cluster mean, empty cluster -> NaN
//   fit / predict      kmeans.py:200-272
//
// Layout: X is (d, N) d-major, so every coordinate row is a unit-stride stream: lanes read
// 16 B (4 consecutive points) per row.  One Lloyd step reads d*4 B per point once and keeps
// everything else (K centroids, K*(d+1)+2 accumulators) in LDS.
//
// Kernels of a single-GPU fit with d = 6, K <= 32 (the anchor clustering):
//   farthest-first   kmeans_init_step_kernel        one launch per centroid; steps >= 2 skip the coordinate read of
//                                                    points that provably keep their running maximum, steps >= 3 whole
//                                                    tiles of 256 points on a 16-byte summary
//   iteration 0      kmeans_assign_kernel body      exact scan + full accumulation (inside the loop kernel's launch); for
//                                                    shards >= 2^17 points it also writes the packed copy (pack_quad)
//   iterations >= 1  kmeans_assign_filter_kernel    f16 MFMA upper bounds + exact certification of the old label,
//                                                    undecided points through an LDS queue; deltas leave as atomics
//                    kmeans_lloyd_chain_kernel      ... one launch per iteration: every workgroup applies the PREVIOUS
//                                                    iteration's update itself (fold of the delta table of exact totals,
//                                                    means, error, convergence) and then assigns; no serial section.
//                                                    Trace-less fits of big shards: packed_assign_body, the same test on
//                                                    an f16 copy of the points (14 B per point instead of 24); its
//                                                    per-launch tables are made by idle wavefronts beside the update's
//                                                    reductions (packed_tables_side; round 4, with the other prologue
//                                                    trims: profiles/r04k_lloyd_launch_stamps.txt)
//                    kmeans_lloyd_persist_kernel    all iterations in one launch (grid barrier without cache fences):
//                                                    shards <= 32768 points and the side-by-side fits of a batch
//   after the loop   kmeans_inertia_kernel          inertia of the last assignment when no trace was requested
// Everything else (other d / K, given labels, the sharded step API) runs the exact scan and a two-kernel fold + update.
//
// Exactness: the reference sums per-cluster coordinates in fp32 in torch's reduction order,
// which a parallel machine cannot reproduce.  Here every coordinate is converted to a 64-bit
// fixed-point integer (truncation, power-of-two scale => exact) and integers are summed, so the
// result is independent of the order: the same bits for any workgroup schedule, grid size or
// number of GPUs, and identical to the CPU oracle (oracle/et_oracle.c).
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "et_common.h"
#include <sched.h>

#include "et_hostring.h"
#include "et_options.h"
#include "et_mfma_filter.h"

#include "et_kmeans_core.inl"  // shared definitions
#include "et_kmeans_filter.inl"  // the matrix-core label filter on fp32 rows (filter_assign_body)
#include "et_kmeans_packed.inl"  // trace-less Lloyd iterations on the packed f16 copy of the points
#include "et_kmeans_chain.inl"  // one launch per iteration
#include "et_kmeans_persist.inl"  // all iterations of a fit in one launch (kmeans_lloyd_persist_kernel), the inertia pass, pre
#include "et_kmeans_init.inl"  // farthest-first initialisation kernels (kmeans.py
#include "et_kmeans_host.inl"  // host side
#include "et_kmeans_loops.inl"  // host side
